# split-f16 projection vs oracle: step-wise token parity on C1-like and C2-like shapes
import sys, time, numpy as np, torch
import os; R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import textgan_pytorch_amd as tg
from textgan_pytorch_amd import synth
from oracle import oracle
import _cases
DEV = torch.device("cuda")
def t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for (V, L, B, R, std) in [(5000, 20, 64, 4, 1.0), (5255, 51, 64, 1, 1.0), (5000, 20, 64, 4, 0.1), (777, 20, 37, 3, 1.0)]:
    gsd = synth.gen_state(V, 32, 32, seed=0, std=std)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    sent = synth.tokens(1, (B, L), V)
    u = synth.uniform01(2, (R, L, L, B))
    og = oracle.Gen(gsd)
    o = oracle.gen_rollout(og, sent, R, u)
    for prec in ("fp32", "split_f16"):
        rep = tg.ops.gen_rollout(gen, t(sent), R, t(u), forced=t(o), precision=prec).cpu().numpy()
        mask = _cases.sampled_mask(R, L, B)
        dis = ((rep != o) & mask).sum(); tot = mask.sum()
        free = tg.ops.gen_rollout(gen, t(sent), R, t(u), precision=prec).cpu().numpy()
        rows = (free == o).all(-1).mean()
        print(V, L, B, R, std, prec, "stepwise disagreements %d / %d = %.2e" % (dis, tot, dis / tot), "rows equal %.4f" % rows, flush=True)
        try:
            _cases.check_token_parity(rep, o, u, og, max_check=50)
            print("   parity check OK")
        except AssertionError as e:
            print("   parity check FAILED", str(e)[:200])
