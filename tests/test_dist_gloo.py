"""world_size-2/3 gloo test of the N>1 path: rollout-index sharding + the single all-gather of raw
reward rows + the local reduce.  The per-rank compute is stood in by the CPU oracle (test-only);
the product's sharding (tg.dist.ShardPlan / gather_rows) is what is under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import textgan_pytorch_amd as tg
        from oracle import oracle
        c = _cases.load(name)
        gen, dis = oracle.Gen(_cases.gen_sd(c)), oracle.Dis(_cases.dis_sd(c))
        sent, u, du = _cases.reward_inputs(c)
        R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
        plan = tg.dist.ShardPlan.from_group(R)
        assert (plan.world, plan.rank) == (world, rank)
        r = oracle.get_reward(gen, dis, sent, R, u, col=int(c["current_k"]) + 1, dropout_p=float(c["dropout_p"]),
                              dropout_uniforms=du, i_lo=plan.i_lo, i_hi=plan.i_hi, reduce=False)
        local = torch.from_numpy(r["raw"][plan.i_lo * L:plan.i_hi * L].copy())
        full = tg.dist.gather_rows(local, plan, L)
        assert tuple(full.shape) == (R * L, B)
        reward = full.view(B, L, R).mean(-1).numpy()   # utils/rollout.py:153
        q.put((rank, full.numpy().copy(), reward))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "reward_tiny_peaked"), (2, "reward_tiny_flat"), (3, "reward_tiny_flat")])
def test_sharded_get_reward_matches_single_rank(world, name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = _cases.load(name)
    for rank, full, reward in got:
        # every rank holds the same full matrix and the same reduced rewards as the 1-rank reference run
        np.testing.assert_allclose(full, c["raw"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(reward, c["reward"], atol=1e-4, rtol=0)
    assert np.array_equal(got[0][1], got[1][1])
