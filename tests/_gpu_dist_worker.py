"""Worker processes of tests/test_y_gpu_dist.py (one process per rank, all on cuda:0 of the one-GPU box).

  nccl1 <port>          world-size-1 "nccl" (RCCL) group created before any other GPU work: ROLLOUT.get_reward through
                        the grouped path == the ungrouped call bit for bit; tg_allgather_rows on a 1-rank ncclComm_t
                        built with ctypes on the RCCL copy torch has loaded.
  gloo2 <port> <rank>   two ranks over gloo sharing the GPU: the product's sharded get_reward (weight broadcast, shard
                        compute on the HIP kernels, all-gather of CUDA rows, reduce) == the single-rank call on the
                        same global uniforms, on both ranks, although rank 1 starts from drifted weights.
Prints "WORKER OK" on success; any assertion ends the process with a non-zero status.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _cases  # noqa: E402
import textgan_pytorch_amd as tg  # noqa: E402

DEV = torch.device("cuda", 0)


def _models(c):
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(int(c["E"]), int(c["H"]), int(c["V"]), int(c["L"])),
                                      _cases.gen_sd(c), DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, int(c["V"])), _cases.dis_sd(c), DEV)
    return gen, dis


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def nccl1(port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)   # before any other GPU call
    c = _cases.load("reward_c1_r2")
    gen, dis = _models(c)
    dis.train()
    sent, u, du = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    du = tg.synth.uniform01(77, (R * L, B, dis.feature_dim)) if du is None else du
    ro = tg.ROLLOUT(gen)
    plain = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), dropout_uniforms=_t(du), return_parts=True)
    ro.group = tg.dist.WORLD
    plan = tg.dist.ShardPlan.from_group(R, ro.group)
    assert (plan.world, plan.rank, plan.i_lo, plan.i_hi) == (1, 0, 0, R)
    grouped = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), dropout_uniforms=_t(du), return_parts=True)
    for a, b in zip(plain, grouped):
        assert torch.equal(a, b)
    # the collectives themselves on RCCL, one rank: all_gather_into_tensor and the weight broadcast's primitive
    full = torch.empty_like(plain[1])
    dist.all_gather_into_tensor(full, plain[1].contiguous())
    assert torch.equal(full, plain[1])
    flat = gen.lstm2out.weight.detach().reshape(-1).clone()
    dist.broadcast(flat, src=0)
    assert torch.equal(flat, gen.lstm2out.weight.detach().reshape(-1))
    assert tg.dist.sync_weights(gen, dis, tg.dist.WORLD) == 0            # one rank: nothing to send

    # tg_allgather_rows for real: a 1-rank communicator from the RCCL copy torch loaded, made with ctypes
    rccl = None
    for line in open("/proc/self/maps"):
        path = line.split()[-1]
        if os.path.basename(path).startswith("librccl.so"):
            rccl = C.CDLL(path)
            break
    assert rccl is not None, "torch's RCCL is not mapped into the process"
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0 and comm.value
    raw = plain[1].clone()
    want = raw.clone()
    lib = tg._lib.lib()
    stream = tg._lib.stream_ptr()
    rc = lib.tg_allgather_rows(raw.data_ptr(), raw.shape[0], raw.shape[1], comm, stream)
    assert rc == 0, lib.tg_last_error()
    torch.cuda.synchronize()
    assert torch.equal(raw, want)                                        # world 1, in place: every row is rank 0's
    red = tg.ops.reward_reduce(raw, B, L, R)
    assert torch.equal(red, plain[0])
    assert lib.tg_allgather_rows(raw.data_ptr(), 0, raw.shape[1], comm, stream) == -1
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclCommDestroy(comm) == 0
    dist.destroy_process_group()
    print("WORKER OK")


def gloo2(port, rank):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    c = _cases.load("reward_c1_r2")
    gen, dis = _models(c)
    dis.eval()
    sent, u, _ = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    single = tg.ROLLOUT(gen).get_reward(_t(sent), R, dis, uniforms=_t(u), return_parts=True)   # before any drift
    if rank == 1:                        # this replica's own optimizer steps: weights drifted from rank 0's
        with torch.no_grad():
            for p in list(gen.parameters()) + list(dis.parameters()):
                p.mul_(1.01)
    ro = tg.ROLLOUT(gen)
    ro.group = tg.dist.WORLD
    rew, raw, comp = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), return_parts=True)
    plan = tg.dist.ShardPlan.from_group(R, ro.group)
    assert (plan.world, plan.rank) == (2, rank) and comp.shape[0] == plan.i_hi - plan.i_lo
    assert torch.equal(comp, single[2][plan.i_lo:plan.i_hi])             # this rank's shard of the completions
    assert torch.equal(raw, single[1]) and torch.equal(rew, single[0])   # full rows and rewards on both ranks
    # (the single-rank call itself is held to this golden in test_gpu_parity.test_get_reward)
    # without the broadcast the drifted rank would have produced other rows
    if rank == 1:
        assert np.array_equal(gen.lstm2out.weight.detach().cpu().numpy(), _cases.gen_sd(c)["lstm2out.weight"])
    print("rank %d: sharded call equals the single-rank call" % rank, flush=True)
    # uneven and empty shards: rollout_num 3 over two ranks (2 + 1) and rollout_num 1 (rank 1 owns nothing)
    for R2 in (3, 1):
        u2 = _t(tg.synth.uniform01(500 + R2, (R2, L, L, B)))
        ro1 = tg.ROLLOUT(gen)
        want = ro1.get_reward(_t(sent), R2, dis, uniforms=u2, return_parts=True)
        got = ro.get_reward(_t(sent), R2, dis, uniforms=u2, return_parts=True)
        p2 = tg.dist.ShardPlan.from_group(R2, ro.group)
        assert got[2].shape[0] == p2.i_hi - p2.i_lo and torch.equal(got[2], want[2][p2.i_lo:p2.i_hi])
        assert torch.equal(got[1], want[1]) and torch.equal(got[0], want[0])
    print("rank %d: uneven and empty shards equal the single-rank call" % rank, flush=True)
    # LeakGAN's reward shards the same way (rollout_num 2 of the golden, one rollout index per rank)
    lc = _cases.load("leak_tiny")
    lgsd, ldsd, lsent, lu, _ = _cases.leak_inputs(lc)
    lgen = tg.modules.load_numpy_state(
        tg.modules.LeakGAN_G(int(lc["E"]), int(lc["H"]), int(lc["V"]), int(lc["L"]), goal_size=int(lc["K"]),
                             step_size=int(lc["step"]), batch_size=int(lc["B"])), lgsd, DEV)
    ldis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, int(lc["V"])), ldsd, DEV).eval()
    lwant = tg.ROLLOUT(lgen).get_reward_leakgan(_t(lsent), int(lc["R"]), ldis, 0, uniforms=_t(lu), return_parts=True)
    lro = tg.ROLLOUT(lgen)
    lro.group = tg.dist.WORLD
    lgot = lro.get_reward_leakgan(_t(lsent), int(lc["R"]), ldis, 0, uniforms=_t(lu), return_parts=True)
    lp = tg.dist.ShardPlan.from_group(int(lc["R"]), lro.group)
    assert torch.equal(lgot[2], lwant[2][lp.i_lo:lp.i_hi])
    assert torch.equal(lgot[1], lwant[1]) and torch.equal(lgot[0], lwant[0])
    print("rank %d: sharded get_reward_leakgan equals the single-rank call" % rank, flush=True)
    # no explicit uniforms: private per-rank streams, so the two halves are different rollouts, not copies
    rew2, raw2, comp2 = ro.get_reward(_t(sent), R, dis, return_parts=True)
    mine = comp2[0, 0].cpu().contiguous()
    both = [torch.empty_like(mine) for _ in range(2)]
    dist.all_gather(both, mine)
    assert not torch.equal(both[0], both[1])
    assert bool(((raw2 >= 0) & (raw2 <= 1)).all()) and tuple(rew2.shape) == (B, L)
    print("rank %d: private streams differ" % rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("WORKER OK")


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(75, exit=True)   # a stuck collective names its line instead of hanging the suite
    if sys.argv[1] == "nccl1":
        nccl1(int(sys.argv[2]))
    else:
        gloo2(int(sys.argv[2]), int(sys.argv[3]))
