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
        # sharding is opt-in: an initialised default group alone changes nothing
        assert tg.dist.ShardPlan.from_group(R).world == 1 and tg.dist.ShardPlan.from_group(R, None).rank == 0
        plan = tg.dist.ShardPlan.from_group(R, tg.dist.WORLD)
        assert (plan.world, plan.rank) == (world, rank)
        r = oracle.get_reward(gen, dis, sent, R, u, col=int(c["current_k"]) + 1, dropout_p=float(c["dropout_p"]),
                              dropout_uniforms=du, i_lo=plan.i_lo, i_hi=plan.i_hi, reduce=False)
        local = torch.from_numpy(r["raw"][plan.i_lo * L:plan.i_hi * L].copy())
        full = tg.dist.gather_rows(local, plan, L, tg.dist.WORLD)
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


def _sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import textgan_pytorch_amd as tg
        from oracle import oracle
        c = _cases.load("reward_tiny_flat")
        R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
        gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(int(c["E"]), int(c["H"]), int(c["V"]), L),
                                          _cases.gen_sd(c), "cpu")
        dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, int(c["V"])), _cases.dis_sd(c), "cpu")
        torch.manual_seed(5)            # the same seed on every rank, as a replicated trainer has it
        if rank != 0:                   # replicas drifted: an optimizer step of their own
            with torch.no_grad():
                for p in list(gen.parameters()) + list(dis.parameters()):
                    p.add_(0.01 * (rank + 1) * torch.randn_like(p))
        before = float(gen.lstm2out.weight.detach().abs().sum())
        # per-module selection (ROLLOUT.sync_weights = "gen"): the generator arrives, the scorer is left as it is
        ro0 = tg.ROLLOUT(gen)
        ro0.group, ro0.sync_weights = tg.dist.WORLD, "gen"
        ro0._sync(dis)
        assert np.array_equal(gen.lstm2out.weight.detach().numpy(), _cases.gen_sd(c)["lstm2out.weight"])
        assert np.array_equal(dis.highway.weight.detach().numpy(), _cases.dis_sd(c)["highway.weight"]) == (rank == 0)
        ro0.sync_weights = "generator"
        with pytest.raises(RuntimeError):
            ro0._sync(dis)
        nbytes = tg.dist.sync_weights(gen, dis, tg.dist.WORLD, src=0)
        want = sum(p.numel() * 4 for m in (gen, dis) for p in m.parameters())
        assert nbytes == want, (nbytes, want)
        assert tg.dist.sync_weights(gen, dis, None) == 0      # no group: nothing to do
        sd_g = {k: v.numpy() for k, v in gen.state_dict().items()}
        sd_d = {k: v.numpy() for k, v in dis.state_dict().items()}
        for k, v in _cases.gen_sd(c).items():                 # rank 0's weights, bit for bit, on every rank
            assert np.array_equal(sd_g[k], v), k
        for k, v in _cases.dis_sd(c).items():
            assert np.array_equal(sd_d[k], v), k
        # the sharded reward call on the synchronised weights (per-rank compute stood in by the oracle)
        sent, u, du = _cases.reward_inputs(c)
        plan = tg.dist.ShardPlan.from_group(R, tg.dist.WORLD)
        r = oracle.get_reward(oracle.Gen(sd_g), oracle.Dis(sd_d), sent, R, u, i_lo=plan.i_lo, i_hi=plan.i_hi,
                              reduce=False)
        full = tg.dist.gather_rows(torch.from_numpy(r["raw"][plan.i_lo * L:plan.i_hi * L].copy()), plan, L,
                                   tg.dist.WORLD)
        # private uniforms of a sharded call: same torch seed on every rank, different draws per rank
        ro = tg.ROLLOUT(gen)
        mine = ro._rand((4, 8), "cpu", plan)
        again = tg.ROLLOUT(gen)._rand((4, 8), "cpu", plan)
        # re-seeding reaches the private streams: another seed gives other draws, the first seed the first
        torch.manual_seed(6)
        other = ro._rand((4, 8), "cpu", plan)
        assert not torch.equal(other, mine)
        torch.manual_seed(5)
        assert torch.equal(ro._rand((4, 8), "cpu", plan), mine)
        ro.generator = torch.Generator().manual_seed(99)
        g1 = ro._rand((4, 8), "cpu", plan)
        ro.generator.manual_seed(99)
        assert torch.equal(ro._rand((4, 8), "cpu", plan), ro._rand((4, 8), "cpu", plan)) is False
        ro.generator = torch.Generator().manual_seed(99)
        assert torch.equal(ro._rand((4, 8), "cpu", plan), g1)
        ro.generator = None
        torch.manual_seed(5)
        single = ro._rand((4, 8), "cpu", tg.dist.ShardPlan(R))
        q.put((rank, before, full.numpy().copy(), mine.numpy().copy(), again.numpy().copy(), single.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_sync_weights_and_private_uniforms():
    """Rank 1 starts from perturbed weights; after the one flattened broadcast both ranks hold rank 0's bits and the
    sharded call returns the single-rank golden.  Without explicit uniforms the ranks of a sharded call draw different
    rows although they share the torch seed (identical draws would make rank k a copy of rank 0)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = _cases.load("reward_tiny_flat")
    assert got[0][1] != got[1][1]                                # the replicas did differ before the broadcast
    for rank, _, full, mine, again, single in got:
        np.testing.assert_allclose(full, c["raw"], atol=1e-4, rtol=0)
        assert np.array_equal(mine, again)                       # repeatable from the seed
    assert np.array_equal(got[0][2], got[1][2])
    assert not np.array_equal(got[0][3], got[1][3])              # private per-rank streams
    assert np.array_equal(got[0][5], got[1][5])                  # unsharded: the default generator, same on both
