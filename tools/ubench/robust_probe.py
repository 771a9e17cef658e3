import sys, os, numpy as np, torch
R_="/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import textgan_pytorch_amd as tg
from textgan_pytorch_amd import synth
from oracle import oracle
import _cases
DEV = torch.device("cuda")
def t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
V, L, B, R = 5000, 20, 64, 2
for std, bscale in [(0.01, 1.0), (4.0, 1.0), (1.0, 50.0), (1e-4, 1e-3)]:
    gsd = synth.gen_state(V, 32, 32, seed=3, std=std)
    gsd = {k: np.array(v) for k, v in gsd.items()}
    gsd["lstm2out.bias"] = (gsd["lstm2out.bias"] * bscale).astype(np.float32)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    sent = synth.tokens(1, (B, L), V); u = synth.uniform01(2, (R, L, L, B))
    og = oracle.Gen(gsd); o = oracle.gen_rollout(og, sent, R, u)
    for prec in ("fp32", "split_f16"):
        rep = tg.ops.gen_rollout(gen, t(sent), R, t(u), forced=t(o), precision=prec).cpu().numpy()
        m = _cases.sampled_mask(R, L, B); dis = ((rep != o) & m).sum()
        ok = "ok"
        try: _cases.check_token_parity(rep, o, u, og, max_check=50)
        except AssertionError as e: ok = "FAIL " + str(e)[:120]
        print("gen std", std, "bias x", bscale, prec, "disagree %d/%d" % (dis, m.sum()), ok, flush=True)
N, Ld = 512, 20
for ws, es in [(1e-3, 1.0), (1e3, 1.0), (1.0, 1e-3), (30.0, 30.0)]:
    sd = {k: np.array(v) for k, v in synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=5).items()}
    for k in sd:
        if k.startswith("convs") and k.endswith("weight"): sd[k] = (sd[k] * ws).astype(np.float32)
    sd["embeddings.weight"] = (sd["embeddings.weight"] * es).astype(np.float32)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), sd, DEV).eval()
    seqs = t(synth.tokens(7, (N, Ld), V))
    a = tg.ops.dis_score(dis, seqs, want_logits=True, want_feature=True, precision="fp32")
    b = tg.ops.dis_score(dis, seqs, want_logits=True, want_feature=True, precision="split_f16")
    od = oracle.dis_forward(oracle.Dis(sd), seqs.cpu().numpy(), col=1, want_feature=True)
    sc = float(a["feature"].abs().max())
    print("dis conv x", ws, "emb x", es, "feat max %.3g" % sc, "split-fp32 rel %.2e" % (float((a["feature"]-b["feature"]).abs().max())/sc),
          "fp32-oracle rel %.2e" % (np.abs(a["feature"].cpu().numpy()-od["feature"]).max()/sc),
          "reward diff split %.2e" % float((a["reward"]-b["reward"]).abs().max()), flush=True)
