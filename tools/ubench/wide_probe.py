"""Step-wise token disagreement of both generator paths with the CPU oracle at the scaled config's shape
(V 32000, E = H = 512), on a batch small enough for the oracle: prints the rate and the CDF gap at each disagreement.

    python tools/ubench/wide_probe.py [B] [L] [R]        (default 64 20 2: 24 320 draws, ~1 TFLOP of oracle work)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases  # noqa: E402
import textgan_pytorch_amd as tg  # noqa: E402
from oracle import oracle  # noqa: E402

B, L, R = [int(x) for x in (sys.argv[1:4] + ["64", "20", "2"][len(sys.argv) - 1:])]
V, E, H = 32000, 512, 512
gsd = tg.synth.gen_state(V, E, H, seed=0, std=0.05)
gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, "cuda")
og = oracle.Gen(gsd)
sent, u = tg.synth.tokens(1, (B, L), V), tg.synth.uniform01(2, (R, L, L, B))
t0 = time.perf_counter()
ref = oracle.gen_rollout(og, sent, R, u)
print("oracle: %.1f s for %d draws on %d threads" % (time.perf_counter() - t0, R * B * L * (L - 1) // 2, oracle.num_threads()))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
m = _cases.sampled_mask(R, L, B)
for prec in ("fp32", "split_f16"):
    rep = tg.ops.gen_rollout(gen, t(sent), R, t(u), forced=t(ref), precision=prec).cpu().numpy()
    diff = (rep != ref) & m
    print("%s: %d disagreements of %d draws (%.2e)" % (prec, int(diff.sum()), int(m.sum()), diff.sum() / m.sum()))
    for (i, gi, b, tt) in np.argwhere(diff)[:10]:
        r = oracle.gen_teacher_forced(og, ref[i, gi, b][None, :], want_logits=True)
        lg = r["logits"][0, tt - 1].astype(np.float64)
        w = np.exp(lg - lg.max())
        cdf = np.cumsum(w) / w.sum()
        a, c = int(rep[i, gi, b, tt]), int(ref[i, gi, b, tt])
        lo, hi = min(a, c), max(a, c)
        print("   (i %d, given_num %d, b %d, t %d): got %d, oracle %d, |cdf - u| = %.2e, p = %.2e"
              % (i, gi + 1, b, tt, a, c, np.abs(cdf[lo:hi] - float(u[i, gi, tt, b])).max(), w[c] / w.sum()))
    free = tg.ops.gen_rollout(gen, t(sent), R, t(u), precision=prec).cpu().numpy()
    print("   free running: %.4f of the rows identical to the oracle's" % (free == ref).all(-1).mean())
