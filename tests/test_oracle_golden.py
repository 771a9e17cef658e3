"""CPU oracle (oracle/textgan_oracle.c) against golden vectors produced by the
real reference (tools/make_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest

import _cases
from oracle import oracle

synth = _cases.synth


@pytest.mark.parametrize("name", _cases.names("get_reward"))
def test_get_reward_matches_reference(name):
    c = _cases.load(name)
    gen, dis = oracle.Gen(_cases.gen_sd(c)), oracle.Dis(_cases.dis_sd(c))
    sent, u, du = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    ref = c["completions"].astype(np.int64)
    # step-wise (replay) parity: north_star >= 99.9 %, disagreements only at fp32 CDF ties
    replay = oracle.gen_rollout(gen, sent, R, u, forced=ref)
    _cases.check_token_parity(replay, ref, u, gen)
    # free-running: ties cascade down a row, so only a loose bound here; rewards on exact rows
    r = oracle.get_reward(gen, dis, sent, R, u, col=int(c["current_k"]) + 1,
                          dropout_p=float(c["dropout_p"]), dropout_uniforms=du)
    same = (r["completions"] == ref).all(-1).reshape(R * L, B)
    assert same.mean() >= 0.98, same.mean()
    np.testing.assert_allclose(r["raw"][same], c["raw"][same], atol=1e-4, rtol=0)
    if same.all():
        np.testing.assert_allclose(r["reward"], c["reward"], atol=1e-4, rtol=0)
    # D alone on the reference's own completions: isolates the scorer from sampling ties
    d = oracle.dis_forward(dis, c["completions"].reshape(R * L * B, L), col=int(c["current_k"]) + 1,
                           dropout_p=float(c["dropout_p"]),
                           dropout_uniforms=None if du is None else du.reshape(R * L * B, -1))
    np.testing.assert_allclose(d["reward"].reshape(R * L, B), c["raw"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(oracle.reward_reduce(c["raw"], B, L, R), c["reward"], atol=1e-6, rtol=0)


@pytest.mark.parametrize("name", _cases.names("forward_nll"))
def test_forward_and_nll_match_reference(name):
    c = _cases.load(name)
    gen = oracle.Gen(_cases.gen_sd(c))
    batch, L = int(c["batch"]), int(c["L"])
    nb = (int(c["N"]) // batch) * batch
    r = oracle.gen_teacher_forced(gen, c["inp"][:nb], c["target"][:nb], want_logits="logits" in c)
    np.testing.assert_allclose(r["logp"], c["logp_target"], rtol=1e-5, atol=1e-5)
    per_batch = -r["logp"].reshape(nb // batch, batch * L).mean(1)
    np.testing.assert_allclose(per_batch, c["per_batch_loss"], rtol=1e-5)
    assert round(float(per_batch.astype(np.float64).mean()), 4) == pytest.approx(c["nll"], abs=1.01e-4)
    if "logits" in c:
        np.testing.assert_allclose(r["logits"][:8], c["logits"], rtol=1e-5, atol=2e-5)
        logprobs = r["logits"][:8] - r["lse"][:8, :, None]
        np.testing.assert_allclose(logprobs, c["logprobs"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name", _cases.names("sample"))
def test_sample_matches_reference(name):
    c = _cases.load(name)
    gen = oracle.Gen(_cases.gen_sd(c))
    ns, batch, L = int(c["num_samples"]), int(c["batch"]), int(c["L"])
    num_batch = ns // batch + 1 if ns != batch else 1
    N = num_batch * batch
    u = synth.uniform01(int(c["useed"]), (L, N))
    forced = np.zeros((N, L), np.int64)
    forced[:ns] = c["samples"]
    out = oracle.gen_sample(gen, N, L, int(c["start_letter"]), u, forced=forced)[:ns]
    assert (out == c["samples"]).mean() >= 0.999
    free = oracle.gen_sample(gen, N, L, int(c["start_letter"]), u)[:ns]
    assert (free == c["samples"]).all(-1).mean() >= 0.98


@pytest.mark.parametrize("name", _cases.names("dis"))
def test_discriminator_matches_reference(name):
    c = _cases.load(name)
    dis = oracle.Dis(_cases.dis_sd(c))
    seqs = synth.tokens(int(c["sseed"]), (int(c["N"]), int(c["L"])), int(c["V"]))
    r = oracle.dis_forward(dis, seqs, want_feature=True)
    np.testing.assert_allclose(r["feature"], c["feature"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["logits"], c["logits"], rtol=1e-5, atol=1e-6)
    assert int(c["short_input_raises"]) == 1
    if max(dis.w.f[k] for k in range(dis.w.n_filters)) > 3:
        with pytest.raises(RuntimeError):
            oracle.dis_forward(dis, seqs[:, :3])


def test_canonical_sampler_properties():
    rng = np.random.default_rng(0)
    for V in (1, 5, 16, 129, 200, 5000, 5255):
        p = rng.random((64, V)).astype(np.float32) ** 4
        p /= p.sum(1, keepdims=True)
        u = rng.random(64).astype(np.float32)
        tok = oracle.sample_from_probs(p, u)
        cdf = np.cumsum(p.astype(np.float64), 1)
        ref = np.minimum((cdf > (u[:, None] * cdf[:, -1:])).argmax(1), V - 1)
        # identical up to fp32 ties: the drawn token's cdf bracket must contain u within eps
        lo = np.where(tok > 0, cdf[np.arange(64), np.maximum(tok - 1, 0)], 0.0)
        hi = cdf[np.arange(64), tok]
        assert np.all(lo - 1e-5 <= u * cdf[:, -1]) and np.all(u * cdf[:, -1] <= hi + 1e-5)
        assert (tok == ref).mean() > 0.95
    # degenerate: u -> 1 picks the last valid entry, one-hot rows pick the hot entry
    p = np.zeros((3, 50), np.float32); p[0, 7] = 1; p[1, 49] = 1; p[2, 0] = 1
    assert oracle.sample_from_probs(p, np.array([0.5, 0.99999994, 0.0], np.float32)).tolist() == [7, 49, 0]


@pytest.mark.parametrize("name", _cases.names("mali_reward"))
def test_mali_reward_matches_reference(name):
    """oracle.mali_reward against MaliGANInstructor.get_mali_reward run from the reference source
    (tools/make_goldens_mali.py).  Values are O(1/B): the tolerance is relative to that scale."""
    c = _cases.load(name)
    V, L, B, R = int(c["V"]), int(c["L"]), int(c["B"]), int(c["R"])
    od = oracle.Dis(synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=int(c["dseed"])))
    samples = synth.tokens(int(c["sseed"]), (B, L), V)
    du = synth.uniform01(int(c["useed"]), (R, B, od.F))
    rew = oracle.mali_reward(od, samples, R, float(c["dropout_p"]), du)
    assert rew.shape == (B, L)
    np.testing.assert_allclose(rew, c["reward"], atol=1e-4 / B, rtol=1e-4)
    assert np.array_equal(rew, np.broadcast_to(rew[:, :1], rew.shape))


@pytest.mark.parametrize("name", _cases.names("pg_loss"))
def test_pg_loss_grads_match_reference(name):
    """oracle.pg_loss_grads (hand-written backward) against loss / .grad of the reference's batchPGLoss / adv_loss."""
    c = _cases.load(name)
    sd, inp, target, reward, senti, weight = _cases.pg_inputs(c)
    loss, grads = oracle.pg_loss_grads(oracle.Gen(sd), inp, target, weight, use_prob=senti,
                                       padding_idx=int(c["padding_idx"]))
    assert loss == pytest.approx(float(c["loss"]), rel=2e-6)
    _cases.assert_grads_close(grads, {k: c["grad." + k] for k in _cases.PG_KEYS}, rel=2e-5, what=name)
    assert not grads["embeddings.weight"][int(c["padding_idx"])].any()


@pytest.mark.parametrize("name", _cases.names("leak_reward"))
def test_leak_reward_matches_reference(name):
    """oracle.leak_rollout / get_reward_leakgan against the reference's own rollout_mc_search_leakgan /
    get_reward_leakgan (utils/rollout.py:57-126,156-184) driving LeakGAN_G.forward and LeakGAN_D
    (tools/make_goldens_leak.py): step-wise token parity with CDF-tie check, raw rows and [B,T] rewards."""
    c = _cases.load(name)
    gsd, dsd, sent, u, du = _cases.leak_inputs(c)
    gen, dis = oracle.Leak(gsd, int(c["step"])), oracle.Dis(dsd)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    T = L // int(c["step"])
    ref = c["completions"].astype(np.int64)
    assert ref.shape == (R, T, B, L)
    replay = oracle.leak_rollout(gen, dis, sent, R, u, forced=ref)
    _cases.check_leak_token_parity(replay, ref, u, gen, dis)
    r = oracle.get_reward_leakgan(gen, dis, sent, R, u, col=int(c["current_k"]) + 1, dropout_p=float(c["dropout_p"]),
                                  dropout_uniforms=du)
    same = (r["completions"] == ref).all(-1).reshape(R * T, B)
    assert same.mean() >= 0.98, same.mean()
    np.testing.assert_allclose(r["raw"][same], c["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, T, R).all(-1)
    np.testing.assert_allclose(r["reward"][ok], c["reward"][ok], atol=1e-4, rtol=0)
    # shards of the rollout index on their own equal the full call
    if R > 1:
        part = oracle.get_reward_leakgan(gen, dis, sent, R, u, col=int(c["current_k"]) + 1,
                                         dropout_p=float(c["dropout_p"]), dropout_uniforms=du, i_lo=1, i_hi=R,
                                         reduce=False)
        assert np.array_equal(part["completions"][1:], r["completions"][1:])
        assert np.array_equal(part["raw"][T:], r["raw"][T:]) and not part["raw"][:T].any()
    np.testing.assert_allclose(oracle.reward_reduce(c["raw"], B, T, R), c["reward"], atol=1e-6, rtol=0)
