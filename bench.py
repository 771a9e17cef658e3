#!/usr/bin/env python3
"""bench.py -- Monte-Carlo rollout completions/s on MI355X (BASELINE.json metric).

A step is one ROLLOUT.get_reward call (weight broadcast when sharded + prefix states + all
completions + CNN scorer with live dropout + exchange + reduce) on BASELINE.json configs[1]
(SeqGAN real-data shape: vocab 5255, len 51, emb/hidden 32, batch 64, rollout_num 16) PER GPU;
with N GPUs the call has rollout_num 16*N sharded over the ranks by rollout index (weak
scaling) and one RCCL all-gather of the raw reward rows.  The same JSON line carries a `scaled`
sub-record: configs[2]'s shape (vocab 32k, len 51, emb/hidden 512, batch 4096) at rollout_num 8
IN TOTAL at every N (strong scaling), with its own roofline and cpu_baseline.  Synthetic
weights / tokens (textgan-pytorch_amd/synth.py), inputs resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def self_launch(argv):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks the way the
    driver does -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py ...`
    -- as a CHILD process, before this process has imported torch or made any GPU call (it never does: it only
    relays), and pass rank 0's one JSON line through.  Returns the exit status to leave with."""
    import socket
    n = _requested_gpus(argv)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host
    env["TG_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: no launcher in the environment, starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr,
          flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in proc.stdout:                                 # stderr of the ranks goes straight through
        txt = raw.decode(errors="replace").strip()
        if txt.startswith("{") and '"metric"' in txt:
            line = txt
        elif txt:
            print(txt, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        print("bench.py: the ranks ended without a JSON line", file=sys.stderr)
    return rc


if __name__ == "__main__" and _requested_gpus(sys.argv[1:]) > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(self_launch(sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ROOT)
import textgan_pytorch_amd as tg  # noqa: E402

WORKLOADS = {
    # name: V, L, E, H, B, R per GPU.  c2 is the default (the metric's config); the others are side runs.
    "c1": dict(V=5000, L=20, E=32, H=32, B=64, R=16, desc="configs[0] SeqGAN oracle synthetic"),
    "c2": dict(V=5255, L=51, E=32, H=32, B=64, R=16, desc="configs[1] SeqGAN real-data shape (EMNLP-News-like)"),
    "c3s": dict(V=32000, L=51, E=512, H=512, B=4096, R=1, desc="configs[2] scaled rollout, 1/64 slice (rollout_num 1 of 64)"),
    "c3": dict(V=32000, L=51, E=512, H=512, B=4096, R=64, strong=True,
               desc="configs[2] scaled rollout, full size (rollout_num 64 in total, sharded over the ranks)"),
    "c3r8": dict(V=32000, L=51, E=512, H=512, B=4096, R=8, strong=True,
                 desc="configs[2] scaled rollout shape at rollout_num 8 of 64 (the same total at every --gpus: strong scaling)"),
    "c4": dict(V=5000, L=20, E=32, H=32, B=64, R=16, k=4, dis="sentigan",
               desc="configs[3] SentiGAN k=4 generators, shared 5-way CNN_D, 4 sequential get_reward calls"),
}
SCALED = "c3r8"                 # the `scaled` sub-record of the default line
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md, Peak BF16/FP16 MFMA, dense
# what v_mfma_f32_32x32x16_f16 sustains on the whole chip from registers alone with non-zero operands (no LDS, no
# memory): tools/ubench/mfma_f16_peak.hip, profiles/r02_mfma_f16_sustained.txt -- 1 655 TFLOP/s (the clock settles
# near 1.6 GHz under that power draw); all-zero operands reach the nominal 2 476.  `peak` stays the nominal figure.
SUSTAINED_F16_MFMA_TFLOPS = 1655.0


def flops_model(V, L, E, H, filt, Ed=64, n_out=2):
    """Algorithmic FLOPs.  conv_fresh_per_bi is what the prefix-reuse scorer executes for
    one (rollout i, sentence b) pair, i.e. summed over given_num g = 1..L: only windows t > g - f touch sampled
    tokens; the others come from the per-sentence running-max table."""
    fs, ns = filt
    F = sum(ns)
    conv_per_seq = 2 * Ed * sum(n * f * (L - f + 1) for f, n in zip(fs, ns))
    conv_fresh_per_bi = 2 * Ed * sum(n * f * sum((L - f + 1) - max(0, g - f + 1) for g in range(1, L + 1))
                                     for f, n in zip(fs, ns))
    hw_per_seq = 2 * F * F + 2 * F * n_out
    gen_per_token = 2 * H * V + 2 * 4 * H * (E + H)
    return conv_per_seq, conv_fresh_per_bi, hw_per_seq, gen_per_token


PROFILE_NAME = {"gen_sample": "k_gen<true", "gen_teacher": "k_gen<false",   # kernel symbols in rocprofv3 output
                "genw_proj": "k_genw_gemm<true", "genw_cell": "k_genw_gemm<false, 2, 4"}


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
            if name.startswith(PROFILE_NAME.get(kernel, "k_" + kernel)):
                best = v.get("hbm_bytes_per_launch")
    return best


def cpu_baseline(wl, budget_s=15.0):
    """The CPU oracle (oracle/, the pinned restatement of the reference) timed on this host's cores on a
    bounded sample of the same workload: same shapes, rollout_num reduced to fit ~budget_s."""
    from oracle import oracle
    V, L, B = wl["V"], wl["L"], wl["B"]
    # bound the sample for the scaled config: fewer sentences (the path is row-parallel; the rate per completion
    # does not depend on the batch), sized for ~budget_s at an assumed 50 GFLOP/s of host arithmetic
    per_completion = (2 * wl["H"] * V + 8 * wl["H"] * (wl["E"] + wl["H"])) * (L - 1) / 2 + 80e6
    B = int(min(B, max(2, budget_s * 5e10 / (L * per_completion))))
    gsd = tg.synth.gen_state(V, wl["E"], wl["H"], seed=0, std=1.0 if wl["H"] == 32 else 0.05)
    if wl.get("dis") == "sentigan":
        dsd = tg.synth.dis_state(V, 64, *tg.synth.SENTIGAN_D_FILTERS, n_out=5, seed=0)
    else:
        dsd = tg.synth.dis_state(V, 64, *tg.synth.SEQGAN_D_FILTERS, n_out=2, seed=0)
    og, od = oracle.Gen(gsd), oracle.Dis(dsd)
    sent = tg.synth.tokens(1, (B, L), V)

    def run(R):
        u = tg.synth.uniform01(2, (R, L, L, B))
        du = tg.synth.uniform01(3, (R * L * B, od.F))
        t0 = time.perf_counter()
        oracle.get_reward(og, od, sent, R, u, dropout_p=0.25, dropout_uniforms=du)  # one generator's call
        return time.perf_counter() - t0

    t1 = run(1)
    R = int(max(1, min(wl["R"], budget_s / max(t1, 1e-3))))
    t = run(R) if R > 1 else t1
    return {"value": R * L * B / t, "unit": "completions/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "same workload at batch %d, rollout_num %d (%d completions, %.1f s)" % (B, R, R * L * B, t)}


def bench_nll(args, dev):
    """Side workload c5 (BASELINE configs[4]): NLL.cal_nll over 99 968 sequences (100 000 with drop_last at batch 64),
    V5000 L20 E=H=32 -- one teacher-forced launch (tg_gen_nll) in --gen-precision (default: the library's,
    split_f16).  Not the headline metric: prints sequences/s."""
    V, L, E, H, N, batch = 5000, 20, 32, 32, 99968, 64
    gen = tg.modules.synthetic_generator(V, E, H, L, seed=0, std=1.0, device=dev)
    samples = torch.from_numpy(tg.synth.tokens(5, (N, L), V)).to(dev)
    for _ in range(args.warmup):
        tg.nll.nll_of_samples(gen, samples, batch, precision=args.gen_precision)
    torch.cuda.synchronize()
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nll, _pb = tg.nll.nll_of_samples(gen, samples, batch, precision=args.gen_precision)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = tg._lib.profile_read()["gen_teacher"]
    tg._lib.profile_enable(False)
    flops = (2 * H * V + 8 * H * (E + H)) * N * L
    ach = flops / (ms / n * 1e-3) / 1e12
    split = (args.gen_precision or tg._lib.default_gen_precision) == "split_f16"
    peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    out = {"metric": "NLL sequences/sec", "value": N * args.steps / dt, "unit": "sequences/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f16x3 (split fp32)" if split else "f32",
           "data": "synthetic",
           "config": {"workload": "configs[4] NLL_gen over 99968 sequences: vocab 5000, len 20, emb/hidden 32, batch 64",
                      "tokens_per_step": N * L, "nll": nll},
           "roofline": {"kernel": "gen_teacher", "bound": "mfma", "achieved": ach, "peak": peak,
                        "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                        "mfma_products_per_flop": 3 if split else 1, "issued_frac": ach * (3 if split else 1) / peak,
                        "flops_per_launch": flops, "avg_launch_ms": ms / n}}
    if not args.no_cpu_baseline:
        from oracle import oracle
        og = oracle.Gen(tg.synth.gen_state(V, E, H, seed=0, std=1.0))
        ns = 8192
        s = samples[:ns].cpu().numpy()
        inp = np.concatenate([np.ones((ns, 1), np.int64), s[:, :-1]], 1)
        t0 = time.perf_counter()
        oracle.gen_teacher_forced(og, inp, s)
        t = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": ns / t, "unit": "sequences/s", "cores": oracle.num_threads(), "kind": "port",
                               "sample": "first %d sequences (%.1f s)" % (ns, t)}
    else:
        out["cpu_baseline"] = None
    return out


def bench_pg(args, dev, wide=False):
    """Side workloads pg / pgw: the adversarial step's policy-gradient loss, forward + backward --
    SeqGAN_G.batchPGLoss + loss.backward() of instructor/real_data/seqgan_instructor.py:109-110 (the optimizer's own
    update is torch's and is not timed) -- at the real-data generator shape (pg: V5255 L51 E=H=32 B64, the fused
    kernels) or at configs[2]'s generator shape (pgw: V32000 L51 E=H=512 B64, the GEMM formulation, --gen-precision
    selecting split-fp16 x3 or exact fp32 vocabulary products).  Beside it, on the same GPU, the reference's
    formulation in eager torch (nn.LSTM + Linear + log_softmax + one_hot, what models/SeqGAN_G.py:21-39 executes)."""
    if wide:
        return bench_pgw(args, dev)
    V, L, E, H, B = 5255, 51, 32, 32, 64
    gen = tg.modules.synthetic_generator(V, E, H, L, seed=0, std=0.1, device=dev)
    gen.name = "seqgan"
    target = torch.from_numpy(tg.synth.tokens(31, (B, L), V)).to(dev)
    inp = torch.cat([torch.ones(B, 1, dtype=torch.int64, device=dev), target[:, :-1]], 1)
    reward = torch.from_numpy(tg.synth.uniform01(32, (B, L)).astype(np.float32)).to(dev)

    def step():
        loss = tg.pgloss.batchPGLoss(gen, inp, target, reward)
        gen.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def eager_step():
        h0 = torch.zeros(1, B, H, device=dev)
        out, _ = gen.lstm(gen.embeddings(inp), (h0, h0.clone()))
        logp = torch.log_softmax(gen.lstm2out(out.contiguous().view(-1, H)), -1).view(B, L, V)
        onehot = torch.nn.functional.one_hot(target, V).float()
        loss = -torch.sum(torch.sum(logp * onehot, -1) * reward)
        gen.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def timed(fn, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    for _ in range(max(args.warmup, 1)):
        step()
        eager_step()
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    dt = timed(step, args.steps)
    prof = tg._lib.profile_read()
    tg._lib.profile_enable(False)
    dt_eager = timed(eager_step, args.steps)
    fwd_ms = prof["gen_teacher"][0] / max(prof["gen_teacher"][1], 1) + \
        prof["pg_forward"][0] / max(prof["pg_forward"][1], 1)
    bwd_ms = prof["pg_backward"][0] / max(prof["pg_backward"][1], 1)
    # backward: logits recomputed twice (row sweep, vocabulary sweep) + dh_proj + dW_out: 4 products of 2*H*V per
    # token, plus 5 products of the cell's size through time
    flops = (8 * H * V + 5 * 8 * H * (E + H)) * B * L
    ach = flops / (bwd_ms * 1e-3) / 1e12
    out = {"metric": "policy-gradient step sequences/sec", "value": B / dt, "unit": "sequences/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "batchPGLoss forward + backward: vocab 5255, len 51, emb/hidden 32, batch 64",
                      "tokens_per_step": B * L},
           "kernels": {"pg_forward (states + rows)": {"avg_ms": fwd_ms}, "pg_backward": {"avg_ms": bwd_ms}},
           "torch_eager_same_gpu": {"ms_per_step": dt_eager * 1e3,
                                    "what": "nn.LSTM + Linear + log_softmax + one_hot, forward + backward"},
           "roofline": {"kernel": "pg_backward", "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                        "flops_per_launch": flops, "avg_launch_ms": bwd_ms,
                        "note": "3 264 token rows: a launch- and latency-bound size, far from the MFMA roof"}}
    if not args.no_cpu_baseline:
        from oracle import oracle
        og = oracle.Gen(tg.synth.gen_state(V, E, H, seed=0, std=0.1))
        a = [x.cpu().numpy() for x in (inp, target, reward)]
        t0 = time.perf_counter()
        oracle.pg_loss_grads(og, *a)
        t = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": B / t, "unit": "sequences/s", "cores": 1, "kind": "port",
                               "sample": "one full step, numpy restatement (%.2f s; numpy's BLAS may use more "
                                         "threads for the matrix products)" % t}
    else:
        out["cpu_baseline"] = None
    return out


def bench_pgw(args, dev):
    V, L, E, H, B = 32000, 51, 512, 512, 64
    gen = tg.modules.synthetic_generator(V, E, H, L, seed=0, std=0.05, device=dev)
    gen.name = "seqgan"
    target = torch.from_numpy(tg.synth.tokens(31, (B, L), V)).to(dev)
    inp = torch.cat([torch.ones(B, 1, dtype=torch.int64, device=dev), target[:, :-1]], 1)
    reward = torch.from_numpy(tg.synth.uniform01(32, (B, L)).astype(np.float32)).to(dev)
    prec = args.gen_precision or tg._lib.default_gen_precision
    split = prec == "split_f16"

    def step():
        loss = tg.pgloss.batchPGLoss(gen, inp, target, reward, precision=prec)
        gen.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def eager_step():
        h0 = torch.zeros(1, B, H, device=dev)
        out, _ = gen.lstm(gen.embeddings(inp), (h0, h0.clone()))
        logp = torch.log_softmax(gen.lstm2out(out.contiguous().view(-1, H)), -1).view(B, L, V)
        onehot = torch.nn.functional.one_hot(target, V).float()
        loss = -torch.sum(torch.sum(logp * onehot, -1) * reward)
        gen.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def timed(fn, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    for _ in range(max(args.warmup, 1)):
        step()
        eager_step()
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    dt = timed(step, args.steps)
    prof = tg._lib.profile_read()
    tg._lib.profile_enable(False)
    dt_eager = timed(eager_step, args.steps)
    per_step = {k: prof[k][0] / args.steps for k in ("pgw_forward", "pgw_backward", "pgw_proj", "pgw_proj_dh", "pgw_proj_dw")}
    # the vocabulary products: logits (forward), logits again, dlogits . W_out, dlogits^T . h (backward): 4 x 2 H V per row
    proj_flops = 4 * 2 * H * V * B * L
    proj_ms = per_step["pgw_proj"] + per_step["pgw_proj_dh"] + per_step["pgw_proj_dw"]
    ach = proj_flops / (proj_ms * 1e-3) / 1e12
    one = 2 * H * V * B * L / 1e9   # GFLOP of one product
    peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    out = {"metric": "policy-gradient step sequences/sec", "value": B / dt, "unit": "sequences/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (split-f16 x3 operands, f32 accumulate: the vocabulary products)" if split else "f32",
           "data": "synthetic",
           "config": {"workload": "batchPGLoss forward + backward at configs[2]'s generator shape: vocab 32000, len 51, "
                                  "emb/hidden 512, batch 64", "tokens_per_step": B * L},
           "kernels": {"pgw_forward (states, log-sum-exp rows)": {"ms_per_step": per_step["pgw_forward"]},
                       "pgw_backward (dlogits, operand fragments, BPTT, weight gradients)": {"ms_per_step": per_step["pgw_backward"]},
                       "vocabulary products (4 per step)": {"ms_per_step": proj_ms, "tflops": ach},
                       "logits = h W_out^T + b (2 per step)": {"ms_per_step": per_step["pgw_proj"],
                                                               "tflops": 2 * one / per_step["pgw_proj"]},
                       "dh_proj = dlogits W_out": {"ms_per_step": per_step["pgw_proj_dh"], "tflops": one / per_step["pgw_proj_dh"]},
                       "dW_out = dlogits^T h": {"ms_per_step": per_step["pgw_proj_dw"], "tflops": one / per_step["pgw_proj_dw"]}},
           "torch_eager_same_gpu": {"ms_per_step": dt_eager * 1e3,
                                    "what": "nn.LSTM + Linear + log_softmax + one_hot, forward + backward"},
           "roofline": {"kernel": "pgw_proj", "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": None, "flops_per_launch": proj_flops / 4,
                        "avg_launch_ms": proj_ms / 4,
                        "mfma_products_per_flop": 3 if split else 1, "issued_frac": ach * (3 if split else 1) / peak,
                        "issued_frac_of_sustained": ach * 3 / SUSTAINED_F16_MFMA_TFLOPS if split else None,
                        "note": "achieved = 4 products x 2 H V flops per token row / the summed time of their launches"}}
    if not args.no_cpu_baseline:
        from oracle import oracle
        og = oracle.Gen(tg.synth.gen_state(V, E, H, seed=0, std=0.05))
        nb = 8                                    # a bounded sample: 8 of the 64 sequences
        a = [x[:nb].cpu().numpy() for x in (inp, target, reward)]
        t0 = time.perf_counter()
        oracle.pg_loss_grads(og, *a)
        t = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nb / t, "unit": "sequences/s", "cores": 1, "kind": "port",
                               "sample": "%d of the %d sequences, numpy restatement (%.1f s; numpy's BLAS may use more "
                                         "threads for the matrix products)" % (nb, B, t)}
    else:
        out["cpu_baseline"] = None
    return out


def bench_leak(args, dev):
    """Side workload leak: ROLLOUT.get_reward_leakgan at run_leakgan.py's configuration (vocab 5000,
    len 20, emb/hidden 32, goal_size 16, step_size 4, batch 64, rollout_num 4; LeakGAN_D with live dropout): 1 280
    completions per call, each position of each completion costing one get_feature pass of the scorer."""
    V, L, E, H, K, step, B, R = 5000, 20, 32, 32, 16, 4, 64, args.leak_rollout_num
    T = L // step
    gen = tg.modules.synthetic_leakgan(V, E, H, L, K, step, B, seed=0, std=0.2, device=dev)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=dev)
    dis.dropout.p = 0.2                       # models/LeakGAN_D.py:16
    if args.eval_mode:
        dis.eval()
    sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
    ro = tg.ROLLOUT(gen, True)
    ro.dis_precision = args.dis_precision
    ro.gen_precision = args.gen_precision
    for _ in range(args.warmup):
        ro.get_reward_leakgan(sent, R, dis, 0)
    torch.cuda.synchronize()
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rew = ro.get_reward_leakgan(sent, R, dis, 0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = tg._lib.profile_read()
    tg._lib.profile_enable(False)
    assert tuple(rew.shape) == (B, T) and bool(torch.isfinite(rew).all())
    N = R * T * B
    conv_f, _, hw_f, _ = flops_model(V, L, E, H, tg.synth.SEQGAN_D_FILTERS)
    split = (args.dis_precision or tg._lib.default_dis_precision) == "split_f16"
    # per call: L - 1 feature passes over all N rows + the final scoring; work2goal for the rows whose next token is drawn
    feat_flops = (conv_f + hw_f) * N * L
    proj_rows = sum(R * B * (L - 1 - max(t * step, 0)) for t in range(T))   # positions pos >= given - 1, pos <= L - 2
    proj_flops = 2 * (K * H + K) * V * proj_rows
    kernels = {k: {"avg_ms": ms / n, "launches": n, "ms_per_call": ms / args.steps} for k, (ms, n) in prof.items() if n}
    conv_ms = kernels["dis_conv"]["ms_per_call"]
    ach = conv_f * N * L / (conv_ms * 1e-3) / 1e12
    peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    kernels["dis_conv"]["tflops"] = ach
    kernels["leak_proj"]["tflops"] = proj_flops / (kernels["leak_proj"]["ms_per_call"] * 1e-3) / 1e12
    gsplit = (args.gen_precision or tg._lib.default_gen_precision) == "split_f16"
    kernels["leak_proj"]["peak_tflops"] = PEAK_F16_MFMA_TFLOPS if gsplit else PEAK_FP32_MFMA_TFLOPS
    kernels["leak_proj"]["mfma_products_per_flop"] = 3 if gsplit else 1
    out = {"metric": "LeakGAN MC rollout completions/sec", "value": N * args.steps / dt, "unit": "completions/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (generator: %s work2goal projection, fp32 cells, goal algebra f64; scorer: %s)"
                    % ("split-f16 x3" if gsplit else "fp32 MFMA", "split-f16 x3" if split else "f32"),
           "data": "synthetic",
           "config": {"workload": "run_leakgan.py shape: vocab %d, len %d, emb/hidden %d, goal_size %d, step_size %d, "
                                  "batch %d, rollout_num %d, LeakGAN_D%s" % (V, L, H, K, step, B, R,
                                                                            " eval mode" if args.eval_mode else " with live dropout"),
                      "completions_per_step": N, "feature_passes_per_step": N * (L - 1)},
           "roofline": {"kernel": "dis_conv", "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": None, "flops_per_launch": conv_f * N,
                        "avg_launch_ms": kernels["dis_conv"]["avg_ms"],
                        "note": "flops as the reference spends them: every position re-scores the whole zero-padded row "
                                "of every completion.  The library evaluates the rows that differ from their "
                                "sentence's leader and the windows that hold a token (DESIGN.md 3.7), and the "
                                "kernel times are summed over streams that overlap"},
           "flops_per_step": {"scorer_features": feat_flops, "work2goal_projection": proj_flops},
           "kernels": kernels}
    if not args.no_cpu_baseline:
        from oracle import oracle
        og = oracle.Leak(tg.synth.leak_state(V, E, H, K, 1720, B, seed=0, std=0.2), step)
        od = oracle.Dis(tg.synth.dis_state(V, 64, *tg.synth.SEQGAN_D_FILTERS, n_out=2, seed=0))
        s = tg.synth.tokens(1, (B, L), V)

        def run(Rc):
            u = tg.synth.uniform01(2, (Rc, T, L, B))
            t0 = time.perf_counter()
            oracle.get_reward_leakgan(og, od, s, Rc, u)
            return time.perf_counter() - t0

        t1 = run(1)                                         # one rollout index: a fraction of a second
        Rc = int(max(1, min(64, 10.0 / max(t1, 1e-3))))     # ... then as many as fill about 10 s
        t = run(Rc) if Rc > 1 else t1
        out["cpu_baseline"] = {"value": Rc * T * B / t, "unit": "completions/s", "cores": oracle.num_threads(), "kind": "port",
                               "sample": "same workload at rollout_num %d (%d completions, %.1f s)" % (Rc, Rc * T * B, t)}
    else:
        out["cpu_baseline"] = None
    return out


def run_rollout(args, name, world, rank, dev, steps, warmup):
    """One completions/s record for WORKLOADS[name]: `warmup` untimed get_reward steps, then exactly `steps` timed ones
    between barrier + synchronize on both sides, max over ranks.  Returns the record on rank 0, None elsewhere."""
    wl = WORKLOADS[name]
    V, L, E, H, B = wl["V"], wl["L"], wl["E"], wl["H"], wl["B"]
    strong = bool(wl.get("strong"))
    if strong and wl["R"] % world:
        raise SystemExit("workload %s shards rollout_num %d over the ranks: --gpus must divide it" % (name, wl["R"]))
    R = wl["R"] if strong else wl["R"] * world  # weak scaling: rollout_num 16 per GPU; strong: fixed total
    r_local = R // world                        # rollout indices one rank's launches process
    k_gens = wl.get("k", 1)
    std = 1.0 if H == 32 else 0.05   # N(0,1) "oracle" weights (generator.py:97-100); small std at H = 512
    gens = [tg.modules.synthetic_generator(V, E, H, L, seed=i, std=std, device=dev) for i in range(k_gens)]
    dis_kind = wl.get("dis", "seqgan")
    dis = tg.modules.synthetic_discriminator(dis_kind, V, seed=0, k_label=4, device=dev)
    if args.eval_mode:
        dis.eval()
    sent = torch.from_numpy(tg.synth.tokens(1, (B, L), V)).to(dev)
    ros = []
    for gi, g in enumerate(gens):
        ro = tg.ROLLOUT(g, True)
        ro.generator = torch.Generator(device=dev)
        ro.generator.manual_seed(1234 + 100 * gi)   # the same seed on every rank: the library derives per-rank streams
        if world > 1:
            ro.group = tg.dist.WORLD                # shard the rollout index over the ranks (opt-in); weights are
            # broadcast from rank 0 at the top of every call (once per step); the k generators of a SentiGAN step share
            # one D that nobody steps between their calls, so only the first call of a step carries it
            ro.sync_weights = True if gi == 0 else "gen"
        ro.dis_precision = args.dis_precision
        ro.gen_precision = args.gen_precision
        if args.dropout_rng:
            ro.dropout_rng = args.dropout_rng
        ros.append(ro)

    def step():
        # SentiGAN (k generators, shared D): generators are visited sequentially, column current_k + 1
        # (instructor/real_data/sentigan_instructor.py:137-147)
        out = None
        for gi, ro in enumerate(ros):
            out = ro.get_reward(sent, R, dis, gi if k_gens > 1 else 0)
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    for ro in ros:
        ro.timers = {}               # per-phase events on the launch stream (two event records per phase and call)
    t0 = time.perf_counter()
    for _ in range(steps):
        rewards = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = tg._lib.profile_read()
    tg._lib.profile_enable(False)
    # per-phase milliseconds per step on this rank's stream: weight broadcast, shard compute (generator + scorer),
    # the all-gather of the raw rows, the view(B,L,R).mean reduce; what is left of the step is host time
    phase_names = ("sync_weights", "gen_rollout", "dis_score", "exchange", "reduce")
    phase_ms = [sum(a.elapsed_time(b) for ro in ros for a, b in ro.timers.get(k, [])) / steps for k in phase_names]
    for ro in ros:
        ro.timers = None
    phase_max = list(phase_ms)
    if world > 1:
        t = torch.tensor([dt] + phase_ms, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        phase_max = [float(x) for x in t[1:].tolist()]
    assert tuple(rewards.shape) == (B, L) and bool(torch.isfinite(rewards).all())
    del ros, gens, dis, rewards
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    completions = B * R * L * k_gens  # whole job, per step
    filt = tg.synth.SENTIGAN_D_FILTERS if dis_kind == "sentigan" else tg.synth.SEQGAN_D_FILTERS
    conv_f, conv_fresh_bi, hw_f, gen_f = flops_model(V, L, E, H, filt, n_out=5 if dis_kind == "sentigan" else 2)
    n_local = B * r_local * L  # sequences one rank's launches process per get_reward call
    split = (args.dis_precision or tg._lib.default_dis_precision) == "split_f16"
    gsplit = (args.gen_precision or tg._lib.default_gen_precision) == "split_f16" and E == 32 and H == 32
    gwide = (args.gen_precision or tg._lib.default_gen_precision) == "split_f16" and not (E == 32 and H == 32)
    draws = B * r_local * L * (L - 1) // 2            # sampled tokens of one rank's call
    cell_steps = B * r_local * (L - 1) * (L - 2) // 2  # wide path: rows enter with their prefix state, no cell step
    # flops one rank EXECUTES per get_reward call, per kernel (prefix-reuse conv: fresh windows only; the
    # reference-equivalent conv work, every window of every completion, is reported beside it)
    per_call = {"dis_conv": conv_fresh_bi * B * r_local, "dis_highway": hw_f * n_local,
                "gen_sample": gen_f * draws, "genw_proj": 2 * H * V * draws,
                # the cell GEMM executes the W_hh . h part; W_ih . x comes from a per-call table over the vocabulary
                "genw_cell": 8 * H * H * (cell_steps + B * (L - 1))}
    f16 = {"dis_conv": split, "dis_highway": split, "gen_sample": gsplit, "genw_proj": True, "genw_cell": True}
    peaks = {k: PEAK_F16_MFMA_TFLOPS if v else PEAK_FP32_MFMA_TFLOPS for k, v in f16.items()}
    # split-f16 kernels issue three fp16 MFMA products per algorithmic multiply-add
    issue_factor = {k: 3 if v else 1 for k, v in f16.items()}
    kernels = {}
    calls = steps * k_gens
    for kname, (ms, n) in prof.items():
        if n:
            kernels[kname] = {"avg_ms": ms / n, "launches": n, "ms_per_call": ms / calls}
            if kname in per_call:
                kernels[kname]["tflops"] = per_call[kname] * calls / (ms * 1e-3) / 1e12
                kernels[kname]["peak_tflops"] = peaks[kname]
                kernels[kname]["mfma_products_per_flop"] = issue_factor[kname]
    kernels["dis_conv"]["reference_equivalent_tflops"] = conv_f * n_local / (kernels["dis_conv"]["ms_per_call"] * 1e-3) / 1e12
    per_launch = {k: v * calls / kernels[k]["launches"] for k, v in per_call.items() if k in kernels}
    dom = max(per_launch, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
    ach = kernels[dom]["tflops"]
    out = {
        "metric": "MC rollout completions/sec", "value": completions * steps / dt, "unit": "completions/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32" + (" (split-f16 x3 operands, f32 accumulate: %s)" % ", ".join(
            (["generator projection"] if gsplit else []) + (["generator gates and projection"] if gwide else []) +
            (["scorer conv/highway"] if split else []))
            if (split or gsplit or gwide) else ""), "data": "synthetic",
        "config": {"workload": "%s: vocab %d, len %d, emb/hidden %d, batch %d, rollout_num %d per GPU (%d total), "
                               "%s scorer%s" % (wl["desc"], V, L, H, B, r_local, R, dis_kind,
                                                " eval mode" if args.eval_mode else " with live dropout"),
                   "completions_per_step": completions,
                   "sharding": "rollout index over %d rank(s)%s" % (
                       world, ", weights broadcast from rank 0 once per step, one all-gather of the raw rows" if world > 1 else "")},
        "roofline": {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peaks[dom],
                     "unit": "TFLOP/s", "frac": ach / peaks[dom],
                     "traffic": pmc_traffic(name, dom) if world == 1 else None,
                     "flops_per_launch": per_launch[dom], "avg_launch_ms": kernels[dom]["avg_ms"],
                     # split-f16 kernels issue 3 fp16 MFMA products per algorithmic multiply-add: the share of the
                     # fp16 peak the matrix pipes actually deliver (the kernel is power-limited)
                     "mfma_products_per_flop": issue_factor[dom],
                     "issued_frac": ach * issue_factor[dom] / peaks[dom],
                     # against the measured sustained rate of the same MFMA with real operands (power-limited clock)
                     "issued_frac_of_sustained": (ach * issue_factor[dom] / SUSTAINED_F16_MFMA_TFLOPS) if f16[dom] else None},
        "kernels": kernels,
        # stream time per step and phase (hipEvents around each phase of ROLLOUT.get_reward): rank 0's own and the
        # slowest rank's; `host_and_gaps` = the step's wall time not covered by them on the slowest rank
        "phases": {"unit": "ms per step", "rank0": dict(zip(phase_names, phase_ms)),
                   "max_over_ranks": dict(zip(phase_names, phase_max)),
                   "host_and_gaps": dt / steps * 1e3 - sum(phase_max)},
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(wl)
    elif world == 1:
        out["cpu_baseline"] = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="default", choices=["default"] + sorted(WORKLOADS) + ["c5", "pg", "pgw", "leak"],
                    help="default: the headline line on configs[1] (c2, weak scaling over --gpus) carrying a `scaled` "
                         "sub-record on configs[2]'s shape (c3r8, strong scaling over --gpus); any other name: that "
                         "workload alone")
    ap.add_argument("--scaled-steps", type=int, default=None, help="timed steps of the `scaled` sub-record (default: --steps)")
    ap.add_argument("--scaled-warmup", type=int, default=None, help="warm-up steps of the `scaled` sub-record (default: --warmup)")
    ap.add_argument("--leak-rollout-num", type=int, default=4, help="rollout_num of --workload leak (run_leakgan.py: 4)")
    ap.add_argument("--no-scaled", action="store_true", help="default workload without the `scaled` sub-record")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="dis.eval(): no dropout (the reference leaves it on)")
    ap.add_argument("--dis-precision", default=None, choices=["fp32", "split_f16"],
                    help="scorer conv/highway arithmetic (default: the library default, split_f16)")
    ap.add_argument("--dropout-rng", default=None, choices=["kernel", "torch"],
                    help="scorer dropout masks: counter-based inside the kernel (default) or torch.rand uniforms")
    ap.add_argument("--gen-precision", default=None, choices=["fp32", "split_f16"],
                    help="generator vocabulary-projection arithmetic (default: the library default, split_f16)")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: whatever libraries print there (gloo's connection banner,
    # RCCL or MIOpen notices) goes to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        os.write(json_fd, (json.dumps(record) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start `python bench.py --gpus %d` without a launcher (it starts "
                         "its own ranks) or under torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus, args.gpus))
    # TG_BENCH_REHEARSE=1: every rank on GPU 0 over gloo -- a rehearsal of the N > 1 code path on a one-GPU box (RCCL
    # refuses two ranks on one device); the numbers of such a run mean nothing
    rehearse = os.environ.get("TG_BENCH_REHEARSE") == "1"
    if os.environ.get("TG_BENCH_DRYRUN") == "1":
        # launcher / rendezvous / relay check for machines without a GPU (tests/test_host.py): no measurement at all
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit({"metric": "MC rollout completions/sec", "dry_run": True, "n_gpus": world})
        return
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.workload == "c5":
        if world > 1:
            raise SystemExit("workload c5 (NLL) is a single-GPU side measurement")
        return emit(bench_nll(args, dev))
    if args.workload in ("pg", "pgw"):
        if world > 1:
            raise SystemExit("workload %s (policy-gradient step) is a single-GPU side measurement" % args.workload)
        return emit(bench_pg(args, dev, wide=args.workload == "pgw"))
    if args.workload == "leak":
        if world > 1:
            raise SystemExit("workload leak (LeakGAN rollout) is a single-GPU side measurement")
        return emit(bench_leak(args, dev))
    if args.workload != "default":
        out = run_rollout(args, args.workload, world, rank, dev, args.steps, args.warmup)
    else:
        # the headline: configs[1], rollout_num 16 per GPU (weak scaling) ...
        out = run_rollout(args, "c2", world, rank, dev, args.steps, args.warmup)
        # ... and, in the same line, configs[2]'s shape at a rollout_num one GPU finishes in seconds per step, the same
        # TOTAL work at every --gpus (strong scaling, what BASELINE.json asks of configs[2] at 1/2/4/8 GPUs)
        if not args.no_scaled and WORKLOADS[SCALED]["R"] % world == 0:
            sub = run_rollout(args, SCALED, world, rank, dev,
                              args.steps if args.scaled_steps is None else args.scaled_steps,
                              args.warmup if args.scaled_warmup is None else args.scaled_warmup)
            if rank == 0:
                out["scaled"] = sub
    if rank == 0:
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
