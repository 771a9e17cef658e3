#!/usr/bin/env python3
"""bench.py -- Monte-Carlo rollout completions/s on MI355X (BASELINE.json metric).

A step is one ROLLOUT.get_reward call (prefix states + all completions + CNN scorer with live
dropout + exchange + reduce) on BASELINE.json configs[1] (SeqGAN real-data shape: vocab 5255,
len 51, emb/hidden 32, batch 64, rollout_num 16) PER GPU; with N GPUs the call has
rollout_num 16*N sharded over the ranks by rollout index (weak scaling) and one RCCL
all-gather of the raw reward rows.  Synthetic weights / tokens (textgan-pytorch_amd/synth.py),
inputs resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import textgan_pytorch_amd as tg  # noqa: E402

WORKLOADS = {
    # name: V, L, E, H, B, R per GPU
    "c1": dict(V=5000, L=20, E=32, H=32, B=64, R=16, desc="configs[0] SeqGAN oracle synthetic"),
    "c2": dict(V=5255, L=51, E=32, H=32, B=64, R=16, desc="configs[1] SeqGAN real-data shape (EMNLP-News-like)"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, Peak FP32 (matrix)


def flops_model(V, L, E, H, filt, Ed=64, n_out=2):
    """Algorithmic FLOPs."""
    fs, ns = filt
    F = sum(ns)
    conv_per_seq = 2 * Ed * sum(n * f * (L - f + 1) for f, n in zip(fs, ns))
    hw_per_seq = 2 * F * F + 2 * F * n_out
    gen_per_token = 2 * H * V + 2 * 4 * H * (E + H)
    return conv_per_seq, hw_per_seq, gen_per_token


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json,
    made by tools/summarize_prof.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") != workload:
            continue
        for name, v in d.get("kernels", {}).items():
            if name.startswith("k_" + kernel):
                best = v.get("hbm_bytes_per_launch")
    return best


def cpu_baseline(wl, budget_s=15.0):
    """The CPU oracle (oracle/, the pinned restatement of the reference) timed on this host's cores on a
    bounded sample of the same workload: same shapes, rollout_num reduced to fit ~budget_s."""
    from oracle import oracle
    V, L, B = wl["V"], wl["L"], wl["B"]
    gsd = tg.synth.gen_state(V, wl["E"], wl["H"], seed=0, std=1.0)
    dsd = tg.synth.dis_state(V, 64, *tg.synth.SEQGAN_D_FILTERS, n_out=2, seed=0)
    og, od = oracle.Gen(gsd), oracle.Dis(dsd)
    sent = tg.synth.tokens(1, (B, L), V)

    def run(R):
        u = tg.synth.uniform01(2, (R, L, L, B))
        du = tg.synth.uniform01(3, (R * L * B, od.F))
        t0 = time.perf_counter()
        oracle.get_reward(og, od, sent, R, u, dropout_p=0.25, dropout_uniforms=du)
        return time.perf_counter() - t0

    t1 = run(1)
    R = int(max(1, min(wl["R"], budget_s / max(t1, 1e-3))))
    t = run(R) if R > 1 else t1
    return {"value": R * L * B / t, "unit": "completions/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "same workload at rollout_num %d (%d completions, %.1f s)" % (R, R * L * B, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="dis.eval(): no dropout (the reference leaves it on)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    wl = WORKLOADS[args.workload]
    V, L, E, H, B = wl["V"], wl["L"], wl["E"], wl["H"], wl["B"]
    R = wl["R"] * world  # weak scaling: rollout_num 16 per GPU
    gen = tg.modules.synthetic_generator(V, E, H, L, seed=0, std=1.0, device=dev)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=dev)
    if args.eval_mode:
        dis.eval()
    sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
    ro = tg.ROLLOUT(gen, True)
    ro.generator = torch.Generator(device=dev)
    ro.generator.manual_seed(1234 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ro.get_reward(sent, R, dis)
    barrier()
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rewards = ro.get_reward(sent, R, dis)
    barrier()
    dt = time.perf_counter() - t0
    prof = tg._lib.profile_read()
    tg._lib.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert tuple(rewards.shape) == (B, L) and bool(torch.isfinite(rewards).all())

    if rank == 0:
        completions = B * R * L  # whole job, per step
        conv_f, hw_f, gen_f = flops_model(V, L, E, H, tg.synth.SEQGAN_D_FILTERS)
        n_local = B * wl["R"] * L  # sequences one rank's launches process
        per_launch = {"dis_conv": conv_f * n_local, "dis_highway": hw_f * n_local,
                      "gen_sample": gen_f * B * wl["R"] * L * (L - 1) // 2}
        kernels = {}
        for name, (ms, n) in prof.items():
            if n:
                kernels[name] = {"avg_ms": ms / n, "launches": n}
                if name in per_launch:
                    kernels[name]["tflops"] = per_launch[name] / (ms / n * 1e-3) / 1e12
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
        ach = kernels[dom].get("tflops", 0.0)
        out = {
            "metric": "MC rollout completions/sec", "value": completions * args.steps / dt, "unit": "completions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: vocab %d, len %d, emb/hidden %d, batch %d, rollout_num %d per GPU (%d total), "
                                   "SeqGAN_D scorer%s" % (wl["desc"], V, L, H, B, wl["R"], R,
                                                          " eval mode" if args.eval_mode else " with live dropout"),
                       "completions_per_step": completions, "sharding": "rollout index over %d rank(s)" % world},
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": pmc_traffic(args.workload, dom) if world == 1 else None,
                         "flops_per_launch": per_launch.get(dom), "avg_launch_ms": kernels[dom]["avg_ms"]},
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
