"""Repeat tg_leak_rollout at run_leakgan.py's shape with fixed inputs and compare every call's completions with the
first call's, bit for bit: the call spreads over four streams and reuses feature buffers round-robin, so a missing
dependency would show as a rare difference.  Also interleaves calls with unrelated work on the caller's stream.
    python tools/diag/leak_repeat.py [calls]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import textgan_pytorch_amd as tg  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
V, L, E, H, K, step, B, R = 5000, 20, 32, 32, 16, 4, 64, 4
T = L // step
gen = tg.modules.synthetic_leakgan(V, E, H, L, K, step, B, seed=0, std=0.2, device=dev)
dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=dev).eval()
sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
u = torch.from_numpy(tg.synth.uniform01(2, (R, T, L, B))).to(dev)
given = [t * step + 1 for t in range(T)]
bad = 0
for prec in ("split_f16", "fp32"):
    first = tg.ops.leak_rollout(gen, dis, sent, R, given, u, precision=prec).clone()
    junk = torch.randn(2048, 2048, device=dev)
    for i in range(n):
        if i % 3 == 0:
            junk = junk @ junk * 1e-3          # unrelated work queued in front of the call
        got = tg.ops.leak_rollout(gen, dis, sent, R, given, u, precision=prec)
        if i % 5 == 0:
            junk = junk + 1.0                  # ... and behind it, before the result is read
        if not torch.equal(got, first):
            bad += 1
            print("call %d (%s) differs in %d tokens" % (i, prec, int((got != first).sum())), flush=True)
    torch.cuda.synchronize()
    print("%s: %d calls, %d differ" % (prec, n, bad), flush=True)
sys.exit(1 if bad else 0)
