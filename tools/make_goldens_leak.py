#!/usr/bin/env python3
"""Golden vectors for the LeakGAN rollout from the REAL reference (build container only).

Runs the reference's own ``ROLLOUT.get_reward_leakgan`` / ``rollout_mc_search_leakgan`` (utils/rollout.py:57-126,
156-184) driving ``LeakGAN_G.forward`` (models/LeakGAN_G.py:52-102) and ``LeakGAN_D`` on CPU, with the two
substitutions of tools/make_goldens.py (``torch.multinomial`` -> canonical inverse-CDF draw from pre-drawn uniforms
indexed [i, t, pos, b]; optionally ``dis.dropout`` -> mask-from-uniform).  Stores seeds/shapes of the inputs and the
reference's OUTPUTS only (completions per (i, t), raw reward rows, final [B, T] rewards)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (stubs nltk, puts the reference on sys.path, sets cfg)

import torch  # noqa: E402

cfg = mg.cfg
synth, oracle = mg.synth, mg.oracle
OUT = mg.OUT


def run_leak(name, V, E, H, K, L, B, R, step, gseed, gstd, dseed, sseed, useed, current_k=0, dropout=False):
    from models.LeakGAN_D import LeakGAN_D
    from models.LeakGAN_G import LeakGAN_G
    from utils.rollout import ROLLOUT
    t0 = time.time()
    cfg.batch_size = B
    gen = LeakGAN_G(E, H, V, L, 0, K, step, gpu=False)
    G = gen.goal_out_size
    gen.load_state_dict(mg.to_torch_sd(synth.leak_state(V, E, H, K, G, B, seed=gseed, std=gstd)))
    dis = LeakGAN_D(64, V, 0, gpu=False)
    dis.load_state_dict(mg.to_torch_sd(synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=dseed)))
    T = L // step
    sent = synth.tokens(sseed, (B, L), V)
    u = synth.uniform01(useed, (R, T, L, B))
    slices = [u[i, t, pos] for i in range(R) for t in range(T) for pos in range(t * step + 1, L)]
    p = float(dis.dropout.p)
    du = None
    if dropout:
        du = synth.uniform01(useed + 7, (R * T, B, dis.feature_dim))
        dis.dropout = mg.MaskDropout(p, du)
    else:
        dis.eval()
    comps = []
    orig = ROLLOUT.rollout_mc_search_leakgan

    def spy(self, sentences, d, given_num):
        out = orig(self, sentences, d, given_num)
        comps.append(out.numpy().copy())
        return out

    ROLLOUT.rollout_mc_search_leakgan = spy
    feed = mg.UniformFeed(slices)
    torch.multinomial = feed
    try:
        ro = ROLLOUT(gen, False)
        rew = ro.get_reward_leakgan(torch.from_numpy(sent), R, dis, current_k)
    finally:
        torch.multinomial = mg._real_multinomial
        ROLLOUT.rollout_mc_search_leakgan = orig
    assert feed.c == len(slices)
    comp = np.stack(comps).reshape(R, T, B, L)
    if dropout:
        dis.dropout = mg.MaskDropout(p, du)
    with torch.no_grad():
        raw = np.stack([torch.softmax(dis.forward(torch.from_numpy(comp[i, t])), -1)[:, current_k + 1].numpy()
                        for i in range(R) for t in range(T)])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), kind="leak_reward", V=V, E=E, H=H, K=K, G=G, L=L, B=B, R=R, step=step,
        gseed=gseed, gstd=gstd, dseed=dseed, sseed=sseed, useed=useed, current_k=current_k, dropout=int(dropout),
        dropout_p=p if dropout else 0.0, completions=comp.astype(np.int32), raw=raw.astype(np.float32),
        reward=rew.numpy().astype(np.float32))
    print("%s: %.1fs reward mean %.5f" % (name, time.time() - t0, float(rew.mean())))


def main():
    os.makedirs(OUT, exist_ok=True)
    oracle.build()
    torch.set_num_threads(8)
    only = set(sys.argv[1:])

    def case(name, *a, **k):
        if not only or name in only:
            run_leak(name, *a, **k)

    # name, V, E, H, K, L, B, R, step, gseed, gstd, dseed, sseed, useed
    case("leak_tiny", 120, 32, 32, 16, 20, 6, 2, 4, 201, 0.3, 202, 203, 204)
    case("leak_tiny_dropout", 150, 32, 32, 16, 20, 5, 2, 4, 211, 0.5, 212, 213, 214, dropout=True)
    case("leak_len24_step4", 90, 32, 32, 16, 24, 4, 1, 4, 221, 0.4, 222, 223, 224)
    case("leak_step2", 100, 32, 32, 16, 20, 4, 1, 2, 241, 0.4, 242, 243, 244)    # step_size 2: the goal window quirk
    case("leak_step5_k8", 110, 32, 64, 8, 20, 4, 2, 5, 251, 0.3, 252, 253, 254)   # step 5, goal_size 8, hidden 64
    case("leak_c1_r1", 5000, 32, 32, 16, 20, 64, 1, 4, 231, 0.2, 232, 233, 234)   # run_leakgan.py's shape, rollout_num 1 of 4


if __name__ == "__main__":
    main()
