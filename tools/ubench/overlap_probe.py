"""Does the scorer of one slice of the rollout index overlap usefully with the generator of the next?
configs[1] shape (V5255 L51 E=H=32 B64 rollout_num 16).  Baseline: ROLLOUT.get_reward.  Pipelined: the rollout index
cut into chunks, tg_gen_rollout of chunk c on the current stream, tg_dis_score_rollout of chunk c on a second stream
behind an event -- both entry points already take a slice of the rollout index and give the bits of the whole call.

    python tools/ubench/overlap_probe.py [chunks ...]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import textgan_pytorch_amd as tg  # noqa: E402
from textgan_pytorch_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
V, L, E, H, B, R = 5255, 51, 32, 32, 64, 16
gen = tg.modules.synthetic_generator(V, E, H, L, seed=0, std=1.0, device=dev)
dsd = tg.synth.dis_state(V, 64, *tg.synth.SEQGAN_D_FILTERS, n_out=2, seed=0)
dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), dsd, dev).train()
sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
u = torch.rand(R, L, L, B, device=dev)
ro = tg.ROLLOUT(gen)
side = torch.cuda.Stream()


def pipelined(nchunk, seeds):
    cur = torch.cuda.current_stream()
    comp = torch.empty(R, L, B, L, dtype=torch.int64, device=dev)
    raw = torch.empty(R * L, B, device=dev)
    weights = _lib.dis_weights(dis, None)
    p = float(dis.dropout.p)
    step = (R + nchunk - 1) // nchunk
    side.wait_stream(cur)
    for c, lo in enumerate(range(0, R, step)):
        hi = min(R, lo + step)
        ops.gen_rollout(gen, sent, R, u, i_lo=lo, i_hi=hi, out=comp)
        ev = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.dis_score_rollout(dis, sent, comp[lo:hi], col=1, dropout_p=p, reward_out=raw[lo * L:hi * L],
                                  weights=weights, dropout_seed=seeds[c])
    cur.wait_stream(side)
    return ops.reward_reduce(raw, B, L, R), raw, comp


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


base = timeit(lambda: ro.get_reward(sent, R, dis, uniforms=u))
print("baseline get_reward: %.3f ms" % base, flush=True)
for nchunk in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    seeds = [1000 + c for c in range(nchunk + 1)]
    t = timeit(lambda: pipelined(nchunk, seeds))
    print("pipelined, %d chunks: %.3f ms (%.3f of baseline)" % (nchunk, t, t / base), flush=True)
# same tokens as the unchunked call (the draws do not depend on the slicing)
_, _, comp1 = ro.get_reward(sent, R, dis, uniforms=u, return_parts=True)
_, _, comp2 = pipelined(4, [1, 2, 3, 4, 5])
torch.cuda.synchronize()
print("completions equal:", bool(torch.equal(comp1, comp2)))
