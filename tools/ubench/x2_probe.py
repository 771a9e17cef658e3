"""The 2-product scorer (TG_DIS_SPLIT_F16X2) against the exact-fp32 scorer: largest reward / logit difference on the
reward goldens' completions, on the weight-scale cases of test_scorer_split_precision_across_weight_scales, and on a
discriminator scaled to O(1) logits (what a trained D produces); the 3-product default beside it.

Historical: the "split_f16x2" precision this probe selects was an experimental build that failed the acceptance rule and
is NOT in the library (profiles/r02_x2_scorer_probe.txt holds its output); against the
current library the probe stops at the first split_f16x2 call with "unknown precision".  Kept as the record of how the
numbers were produced."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases, textgan_pytorch_amd as tg
synth = tg.synth
DEV = "cuda"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)

def cmp(dis, seqs, col, what):
    a = tg.ops.dis_score(dis, seqs, col=col, want_logits=True, precision="fp32")
    out = [what]
    for prec in ("split_f16", "split_f16x2"):
        b = tg.ops.dis_score(dis, seqs, col=col, want_logits=True, precision=prec)
        out.append("%s: reward %.2e logit %.2e" % (prec, float((a["reward"] - b["reward"]).abs().max()),
                                                    float((a["logits"] - b["logits"]).abs().max())))
    out.append("(|logit| max %.2f, reward range %.3f..%.3f)" % (float(a["logits"].abs().max()), float(a["reward"].min()), float(a["reward"].max())))
    print("  ".join(out), flush=True)

for name in _cases.names("get_reward"):
    c = _cases.load(name)
    if str(c["dis_kind"]) == "seqgan":
        d = tg.modules.SeqGAN_D(64, int(c["V"]))
    else:
        d = tg.modules.SentiGAN_D(int(c["k_label"]), 64, int(c["V"]))
    dis = tg.modules.load_numpy_state(d, _cases.dis_sd(c), DEV).eval()
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    cmp(dis, t(c["completions"].astype(np.int64).reshape(R * L * B, L)), int(c["current_k"]) + 1, name)

V, L, N = 5000, 20, 4096
for conv_scale, emb_scale, hw_scale, out_scale in [(1e-3, 1, 1, 1), (1e3, 1, 1, 1), (1, 1e-3, 1, 1), (30, 30, 1, 1), (8, 8, 1, 4), (8, 8, 1, 20), (10, 10, 0.3, 60), (20, 20, 0.2, 30)]:
    sd = {k: np.array(v) for k, v in synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=5).items()}
    for k in sd:
        if k.startswith("convs") and k.endswith("weight"):
            sd[k] = (sd[k] * conv_scale).astype(np.float32)
    sd["embeddings.weight"] = (sd["embeddings.weight"] * emb_scale).astype(np.float32)
    sd["highway.weight"] = (sd["highway.weight"] * hw_scale).astype(np.float32)
    sd["feature2out.weight"] = (sd["feature2out.weight"] * out_scale).astype(np.float32)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), sd, DEV).eval()
    cmp(dis, t(synth.tokens(7, (N, L), V)), 1, "scales conv %g emb %g hw %g out %g" % (conv_scale, emb_scale, hw_scale, out_scale))
