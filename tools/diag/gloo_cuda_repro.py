"""Diagnosis of the stalled gloo all-gather of round 2 (gpurun_out/r02f/pytest.log): two ranks on ONE GPU, gloo
backend, CUDA tensors handed to the collectives directly.

    python tools/diag/gloo_cuda_repro.py <stage> [port]      (starts both ranks itself, prints what each one printed)

Stages (each in fresh processes, each rank ends itself after 40 s with the traceback of where it is stuck):
  torch_only   no product code at all: CUDA broadcast of 18 MB, then all_gather_into_tensor of CUDA rows, 3 rounds
  gather_only  all_gather_into_tensor of CUDA rows alone, 3 rounds
  product      the product's sharded get_reward with host staging switched off (TG_DIST_GLOO_DEVICE=1), 3 calls
  product_nosync  the same with ROLLOUT.sync_weights = False (no broadcast before the gather)
  product_sync    product_nosync with torch.cuda.synchronize() in front of the gather: tells "the kernels never finish"
                  from "gloo's copy path stalls"
  product_stream  product_nosync on a side stream instead of the default (null) stream
  torch_busy      torch_only's gather with 50 ms of torch matmuls enqueued in front of it, 6 rounds
  torch_busy_w1   the same in a gloo group of ONE rank (one process on the GPU)
  emulate         no collective on device memory at all: both ranks do by hand what ProcessGroupGloo does with a CUDA
                  tensor (event on the busy current stream, a pool stream waits for it, non-blocking copy to pinned
                  host memory on that stream, a second thread synchronises the stream), then a host-side all_gather
Every step prints a time stamp, so a stall is located between two lines.
"""
import os
import socket
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
T0 = time.time()


def say(rank, msg):
    print("[%6.2f s] rank %d: %s" % (time.time() - T0, rank, msg), flush=True)


def worker(stage, port, rank):
    import faulthandler
    faulthandler.dump_traceback_later(40, exit=True)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    world = 1 if stage == "torch_busy_w1" else 2
    dist.init_process_group("gloo", rank=rank, world_size=world)
    say(rank, "group up")
    if stage == "emulate":
        import threading
        a = torch.randn(4096, 4096, device=dev)
        for it in range(6):
            for _ in range(40):
                a = torch.tanh(a @ a * 1e-2)
            part = torch.full((20, 64), float(rank), device=dev)
            ev = torch.cuda.Event()
            ev.record()
            side = torch.cuda.Stream()
            side.wait_event(ev)
            pinned = torch.empty(20, 64).pin_memory()
            with torch.cuda.stream(side):
                pinned.copy_(part, non_blocking=True)
            say(rank, "round %d: copy enqueued behind the matmuls, a thread synchronises the side stream ..." % it)
            th = threading.Thread(target=side.synchronize)
            th.start()
            th.join()
            say(rank, "round %d: side stream drained (value %.1f), host all_gather ..." % (it, float(pinned[0, 0])))
            both = [torch.empty(20, 64) for _ in range(2)]
            dist.all_gather(both, pinned)
            say(rank, "round %d: done" % it)
    elif stage in ("torch_only", "gather_only", "torch_busy", "torch_busy_w1"):
        a = torch.randn(4096, 4096, device=dev)
        for it in range(6 if stage.startswith("torch_busy") else 3):
            if stage.startswith("torch_busy"):
                for _ in range(40):
                    a = torch.tanh(a @ a * 1e-2)
            if stage == "torch_only":
                flat = torch.full((4_500_000,), float(rank + 1), device=dev)
                torch.cuda.synchronize()
                say(rank, "round %d: broadcast (CUDA tensor) ..." % it)
                dist.broadcast(flat, src=0)
                torch.cuda.synchronize()
                say(rank, "round %d: broadcast done, value %.1f" % (it, float(flat[0])))
            part = torch.full((20, 64), float(rank), device=dev)
            full = torch.empty(20 * world, 64, device=dev)
            say(rank, "round %d: all_gather_into_tensor (CUDA tensors) ..." % it)
            dist.all_gather_into_tensor(full, part)
            torch.cuda.synchronize()
            say(rank, "round %d: gather done, rows %s" % (it, full[::20, 0].tolist()))
    else:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["TG_DIST_GLOO_DEVICE"] = "1"        # hand the CUDA tensors to gloo directly (no host staging)
        import numpy as np
        import _cases
        import textgan_pytorch_amd as tg
        c = _cases.load("reward_c1_r2")
        gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(int(c["E"]), int(c["H"]), int(c["V"]), int(c["L"])),
                                          _cases.gen_sd(c), dev)
        dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, int(c["V"])), _cases.dis_sd(c), dev).eval()
        sent, u, _ = _cases.reward_inputs(c)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        R = int(c["R"])
        single = tg.ROLLOUT(gen).get_reward(t(sent), R, dis, uniforms=t(u))
        torch.cuda.synchronize()
        say(rank, "single-rank call done")
        ro = tg.ROLLOUT(gen)
        ro.group = tg.dist.WORLD
        ro.sync_weights = stage == "product"
        inner = tg.dist._all_gather_into

        def traced(full, part, g):
            if stage == "product_sync":
                say(rank, "  shard kernels enqueued, synchronising ...")
                torch.cuda.synchronize()
                say(rank, "  device idle, entering all_gather_into_tensor ...")
            inner(full, part, g)
            say(rank, "  all_gather_into_tensor returned")
        tg.dist._all_gather_into = traced
        side = torch.cuda.Stream() if stage == "product_stream" else None
        for it in range(6):
            say(rank, "call %d: sharded get_reward ..." % it)
            if side is not None:
                with torch.cuda.stream(side):
                    rew = ro.get_reward(t(sent), R, dis, uniforms=t(u))
            else:
                rew = ro.get_reward(t(sent), R, dis, uniforms=t(u))
            torch.cuda.synchronize()
            say(rank, "call %d: done, equal to single-rank: %s" % (it, bool(torch.equal(rew, single))))
    dist.barrier()
    dist.destroy_process_group()
    say(rank, "WORKER OK")


def main():
    stage = sys.argv[1]
    if len(sys.argv) >= 4:
        return worker(stage, int(sys.argv[2]), int(sys.argv[3]))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), stage, str(port), str(r)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(1 if stage == "torch_busy_w1" else 2)]
    bad = 0
    for r, p in enumerate(procs):
        try:
            out = p.communicate(timeout=90)[0].decode(errors="replace")
        except subprocess.TimeoutExpired:
            p.kill()
            out = p.communicate()[0].decode(errors="replace") + "\n(killed by the parent after 90 s)"
        print("===== stage %s rank %d: exit %s\n%s" % (stage, r, p.returncode, out[-4000:]), flush=True)
        bad |= int(p.returncode != 0)
    return bad


if __name__ == "__main__":
    sys.exit(main())
