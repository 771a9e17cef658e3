"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the real
reference and against the CPU oracle on the same seeded inputs.  Tolerances are the
north_star's: teacher-forced logits / NLL 1e-5 relative (fp32), sampled tokens >= 99.9 %
step-wise with disagreements only at fp32 CDF ties, rewards 1e-4."""
import numpy as np
import pytest
import torch

import _cases
import textgan_pytorch_amd as tg
from oracle import oracle

pytestmark = pytest.mark.gpu
synth = _cases.synth
DEV = "cuda"


def _gen(c):
    g = tg.modules.LSTMGenerator(int(c["E"]), int(c["H"]), int(c["V"]), int(c["L"]))
    return tg.modules.load_numpy_state(g, _cases.gen_sd(c), DEV)


def _dis(c):
    if str(c["dis_kind"]) == "seqgan":
        d = tg.modules.SeqGAN_D(64, int(c["V"]))
    else:
        d = tg.modules.SentiGAN_D(int(c["k_label"]), 64, int(c["V"]))
    return tg.modules.load_numpy_state(d, _cases.dis_sd(c), DEV)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ------------------------------------------------------------------ generator, teacher forced
@pytest.mark.parametrize("name", _cases.names("forward_nll"))
def test_teacher_forced_logits_and_nll(name):
    c = _cases.load(name)
    gen = _gen(c)
    batch, L = int(c["batch"]), int(c["L"])
    nb = (int(c["N"]) // batch) * batch
    inp, tgt = _t(c["inp"][:nb].astype(np.int64)), _t(c["target"][:nb].astype(np.int64))
    want_logits = "logits" in c
    r = tg.ops.gen_teacher_forced(gen, inp, tgt, want_lse=True, want_logits=want_logits, want_states=True)
    logp = r["logp"].cpu().numpy()
    np.testing.assert_allclose(logp, c["logp_target"], rtol=1e-5, atol=1e-5)
    per_batch = -logp.reshape(nb // batch, batch * L).mean(1)
    np.testing.assert_allclose(per_batch, c["per_batch_loss"], rtol=1e-5)
    # oracle on the same inputs: logits, lse, states
    og = oracle.Gen(_cases.gen_sd(c))
    o = oracle.gen_teacher_forced(og, c["inp"][:nb], c["target"][:nb], want_logits=want_logits, want_states=True)
    np.testing.assert_allclose(r["lse"].cpu().numpy(), o["lse"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["states"].cpu().numpy(), o["states"], rtol=1e-4, atol=5e-5)  # internal state; chaotic N(0,1) LSTMs amplify 1-ulp gate differences
    if want_logits:
        lg = r["logits"].cpu().numpy()
        np.testing.assert_allclose(lg[:8], c["logits"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(lg, o["logits"], rtol=1e-5, atol=2e-5)
        lp = tg.generator.forward_logprobs(gen, inp).view(nb, L, -1).cpu().numpy()
        np.testing.assert_allclose(lp[:8], c["logprobs"], rtol=1e-5, atol=2e-5)
    # the drop-in NLL surface: a loader of dicts, and the loader-free form
    loader = [{"input": inp[i:i + batch].cpu(), "target": tgt[i:i + batch].cpu()} for i in range(0, nb, batch)]
    for prec in ("fp32", "split_f16"):  # tg_gen_nll; split: E = H = 32 inside the persistent kernel, other shapes the GEMM path
        assert tg.nll.cal_nll(gen, loader, precision=prec) == pytest.approx(c["nll"], abs=1.01e-4)
        val, pb = tg.nll.nll_of_samples(gen, tgt, batch, start_letter=1, precision=prec)
        assert val == pytest.approx(c["nll"], abs=1.01e-4)
        np.testing.assert_allclose(pb.cpu().numpy(), c["per_batch_loss"], rtol=1e-5)
        lp2, lse2 = tg.ops.gen_nll(gen, inp, tgt, precision=prec, want_lse=True)
        np.testing.assert_allclose(lp2.cpu().numpy(), c["logp_target"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(lse2.cpu().numpy(), o["lse"], rtol=1e-5, atol=1e-5)
        if prec == "fp32" and not want_logits:  # (with logits_out the kernel takes its natural-log path)
            assert torch.equal(lp2, r["logp"])  # the same kernel as tg_gen_teacher_forced


def test_generator_forward_surface():
    """tg.generator.forward mirrors LSTMGenerator.forward's call shapes (generator.py:40-60): [N,T] and one-step [N]
    inputs, need_hidden, the zero state of init_hidden and a CARRIED (h, c): the reference accepts any hidden state.
    Log-probabilities against the reference golden, the returned state against the oracle; a pass cut in two at any
    position, the second half started from the first half's state, reproduces the golden's log-probabilities."""
    c = _cases.load("nll_tiny")
    gen = _gen(c)
    inp = c["inp"][:8]
    pred, (h, cc) = tg.generator.forward(gen, _t(inp), gen.init_hidden(8), need_hidden=True)
    N, L = inp.shape
    assert tuple(pred.shape) == (N * L, int(c["V"])) and tuple(h.shape) == (1, N, gen.hidden_dim) == tuple(cc.shape)
    if "logprobs" in c:
        np.testing.assert_allclose(pred.view(N, L, -1).cpu().numpy(), c["logprobs"], rtol=1e-5, atol=2e-5)
    o = oracle.gen_teacher_forced(oracle.Gen(_cases.gen_sd(c)), inp, want_logits=True, want_states=True)
    np.testing.assert_allclose(h[0].cpu().numpy(), o["states"][L][:, 0], atol=1e-4)
    np.testing.assert_allclose(cc[0].cpu().numpy(), o["states"][L][:, 1], atol=1e-4)
    one = tg.generator.forward(gen, _t(inp[:, 0]))              # 1-D input: a single step
    assert tuple(one.shape) == (N, int(c["V"]))
    np.testing.assert_allclose(one.cpu().numpy(), pred.view(N, L, -1)[:, 0].cpu().numpy(), rtol=1e-6, atol=1e-6)
    # carried state: [0, cut) from zeros, [cut, L) from the state it returns -- the same arithmetic step by step
    full = pred.view(N, L, -1)
    for cut in (1, 7, L - 1):
        p1, hid = tg.generator.forward(gen, _t(inp[:, :cut]), None, need_hidden=True)
        p2, hid2 = tg.generator.forward(gen, _t(inp[:, cut:]), hid, need_hidden=True)
        got = torch.cat([p1.view(N, cut, -1), p2.view(N, L - cut, -1)], 1)
        assert torch.equal(got, full)
        assert torch.equal(hid2[0], h) and torch.equal(hid2[1], cc)
        if "logprobs" in c:
            np.testing.assert_allclose(got.cpu().numpy(), c["logprobs"], rtol=1e-5, atol=2e-5)
    # the step loop of LSTMGenerator.sample / rollout_mc_search written with forward(): token by token, state carried
    hid, steps = None, []
    for t in range(5):
        out, hid = tg.generator.forward(gen, _t(inp[:, t]), hid, need_hidden=True)
        steps.append(out)
    assert torch.equal(torch.stack(steps, 1), full[:, :5])
    # slot 0 of the exported states is the state the pass started from
    st_in = torch.stack([h[0], cc[0]], 1).contiguous()
    r = tg.ops.gen_teacher_forced(gen, _t(inp[:, :3]), want_logp=False, want_states=True, states_in=st_in)
    assert torch.equal(r["states"][0], st_in)
    with pytest.raises(RuntimeError, match="hidden must be"):
        tg.generator.forward(gen, _t(inp), (h[:, :4], cc[:, :4]))


@pytest.mark.parametrize("V,E,H", [(300, 64, 64), (900, 96, 160), (500, 32, 512)])
def test_generator_forward_carried_state_other_shapes(V, E, H):
    """The carried state on the generic persistent kernel (H = 64, 512) and on the GEMM formulation (H = 160): cut in two
    == in one piece, and against the oracle from zeros."""
    N, L = 9, 8
    gsd = synth.gen_state(V, E, H, seed=4, std=0.2)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    inp = synth.tokens(9, (N, L), V)
    full, (h, c) = tg.generator.forward(gen, _t(inp), None, need_hidden=True)
    o = oracle.gen_teacher_forced(oracle.Gen(gsd), inp, want_logits=True, want_states=True)
    lp = o["logits"] - o["lse"][..., None]
    np.testing.assert_allclose(full.view(N, L, V).cpu().numpy(), lp, rtol=1e-5, atol=2e-5)
    p1, hid = tg.generator.forward(gen, _t(inp[:, :3]), None, need_hidden=True)
    p2, hid2 = tg.generator.forward(gen, _t(inp[:, 3:]), hid, need_hidden=True)
    got = torch.cat([p1.view(N, 3, V), p2.view(N, L - 3, V)], 1)
    np.testing.assert_allclose(got.cpu().numpy(), full.view(N, L, V).cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(hid2[0].cpu().numpy(), h.cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got.cpu().numpy(), lp, rtol=1e-5, atol=2e-5)


def test_teacher_forced_ragged_rows():
    # N not a multiple of the 32-row tile, N == 1, and prefix-state mode (no projection)
    V, L = 777, 7
    sd = synth.gen_state(V, 32, 32, seed=5, std=0.5)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), sd, DEV)
    og = oracle.Gen(sd)
    for N in (1, 33, 70):
        inp = synth.tokens(N, (N, L), V)
        tgt = synth.tokens(N + 100, (N, L), V)
        r = tg.ops.gen_teacher_forced(gen, _t(inp), _t(tgt), want_lse=True, want_states=True)
        o = oracle.gen_teacher_forced(og, inp, tgt, want_states=True)
        np.testing.assert_allclose(r["logp"].cpu().numpy(), o["logp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r["states"].cpu().numpy(), o["states"], rtol=1e-4, atol=5e-5)  # internal state; chaotic N(0,1) LSTMs amplify 1-ulp gate differences
        s_only = tg.ops.gen_teacher_forced(gen, _t(inp), want_logp=False, want_states=True)["states"]
        assert torch.equal(s_only, r["states"])


# ------------------------------------------------------------------ generator, sampling
GEN_PRECISIONS = ["fp32", "split_f16"]  # TG_GEN_*: E = H = 32 inside the persistent kernel, other shapes tg_genw.hip


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
@pytest.mark.parametrize("name", _cases.names("get_reward"))
def test_rollout_tokens(name, precision):
    c = _cases.load(name)
    gen = _gen(c)
    sent, u, _ = _cases.reward_inputs(c)
    R = int(c["R"])
    ref = c["completions"].astype(np.int64)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(ref), precision=precision).cpu().numpy()
    og = oracle.Gen(_cases.gen_sd(c))
    _cases.check_token_parity(replay, ref, u, og)
    # against the oracle's own replay as well (same forced tokens)
    orep = oracle.gen_rollout(og, sent, R, u, forced=ref)
    m = _cases.sampled_mask(*ref.shape[:3])
    assert (replay[m] == orep[m]).mean() >= 0.999
    # free running: whole rows identical for the vast majority (a tie cascades down its row)
    free = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), precision=precision).cpu().numpy()
    assert (free == ref).all(-1).mean() >= 0.98


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
@pytest.mark.parametrize("name", _cases.names("sample"))
def test_sample(name, precision):
    c = _cases.load(name)
    gen = _gen(c)
    ns, batch, L = int(c["num_samples"]), int(c["batch"]), int(c["L"])
    num_batch = ns // batch + 1 if ns != batch else 1
    N = num_batch * batch
    u = synth.uniform01(int(c["useed"]), (L, N))
    forced = np.zeros((N, L), np.int64)
    forced[:ns] = c["samples"]
    out = tg.ops.gen_sample(gen, N, L, int(c["start_letter"]), _t(u), forced=_t(forced),
                            precision=precision).cpu().numpy()[:ns]
    # >= 99.9 % step-wise, and every disagreement a CDF tie (the rollout path's rule, north_star)
    _cases.check_sample_token_parity(out, c["samples"].astype(np.int64), u[:, :ns], oracle.Gen(_cases.gen_sd(c)),
                                     int(c["start_letter"]))
    s = tg.generator.sample(gen, ns, batch, int(c["start_letter"]), uniforms=_t(u))
    assert s.device.type == "cpu" and s.dtype == torch.int64 and tuple(s.shape) == (ns, L)
    assert (s.numpy() == c["samples"]).all(-1).mean() >= 0.98


def test_rollout_subranges_and_mc_search():
    c = _cases.load("reward_tiny_flat")
    gen = _gen(c)
    sent, u, _ = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    full = tg.ops.gen_rollout(gen, _t(sent), R, _t(u))
    part = torch.full_like(full, -1)
    tg.ops.gen_rollout(gen, _t(sent), R, _t(u), g_lo=3, g_hi=5, i_lo=1, i_hi=2, out=part)
    assert torch.equal(part[1, 2:5], full[1, 2:5])
    assert (part[0] == -1).all() and (part[1, :2] == -1).all() and (part[1, 5:] == -1).all()
    ro = tg.ROLLOUT(gen)
    one = ro.rollout_mc_search(_t(sent), 4, uniforms=_t(u[:1]))
    assert torch.equal(one, full[0, 3])
    assert torch.equal(full[:, L - 1], _t(sent).expand(R, B, L))  # given_num == L is the sentence itself


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
@pytest.mark.parametrize("V,B,L,R", [(2, 1, 20, 1), (3, 1, 21, 2), (17, 2, 20, 3), (129, 33, 20, 1), (16, 7, 25, 5),
                                     (5255, 3, 51, 1)])
def test_get_reward_edge_shapes_against_oracle(V, B, L, R, precision):
    """The smallest and most ragged calls get_reward accepts at E = H = 32: a two-word vocabulary (one sampler tile,
    mostly padding), a single sentence, a single rollout, batches that are not a multiple of anything, L just over the
    widest filter.  Step-wise token parity and rewards against the oracle."""
    gsd = synth.gen_state(V, 32, 32, seed=V + B, std=1.0)
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=5)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), dsd, DEV)
    og, od = oracle.Gen(gsd), oracle.Dis(dsd)
    sent, u = synth.tokens(31, (B, L), V), synth.uniform01(32, (R, L, L, B))
    du = synth.uniform01(33, (R * L, B, od.F))
    o = oracle.get_reward(og, od, sent, R, u, dropout_p=0.25, dropout_uniforms=du)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(o["completions"]), precision=precision).cpu().numpy()
    _cases.check_token_parity(replay, o["completions"], u, og)
    ro = tg.ROLLOUT(gen)
    ro.gen_precision = precision
    rew, raw, comp = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), dropout_uniforms=_t(du), return_parts=True)
    assert tuple(rew.shape) == (B, L) and tuple(raw.shape) == (R * L, B)
    same = (comp.cpu().numpy() == o["completions"]).all(-1).reshape(R * L, B)
    np.testing.assert_allclose(raw.cpu().numpy()[same], o["raw"][same], atol=1e-4, rtol=0)
    # every output whose R contributing raw entries come from agreeing completions (one cascaded tie hides only its own)
    ok = same.reshape(B, L, R).all(-1)
    assert ok.mean() >= 0.9
    np.testing.assert_allclose(rew.cpu().numpy()[ok], o["reward"][ok], atol=1e-4, rtol=0)
    # given_num == L scores the sentence itself, whatever the uniforms
    assert torch.equal(comp[:, L - 1], _t(sent).expand(R, B, L))


@pytest.mark.parametrize("V,E,H,precision", [(200, 32, 32, "fp32"), (200, 32, 32, "split_f16"), (5255, 32, 32, "fp32"),
                                             (5255, 32, 32, "split_f16"), (900, 64, 64, "fp32"), (900, 64, 96, "split_f16")])
def test_rollout_uniform_extremes(V, E, H, precision):
    """Uniforms at the ends of [0, 1): u = 0 draws the first entry with a non-zero probability, u = 1 - 2^-24 the last
    one the CDF reaches; a mix of both, of exact powers of two and of ordinary values must match the oracle's tokens."""
    B, L, R = 5, 20, 2
    gsd = synth.gen_state(V, E, H, seed=14, std=1.0 if H == 32 else 0.3)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    sent = synth.tokens(51, (B, L), V)
    u = synth.uniform01(52, (R, L, L, B)).copy()
    flat = u.reshape(-1)
    flat[0::7] = 0.0
    flat[1::7] = np.float32(1.0) - np.float32(2.0 ** -24)
    flat[2::7] = 0.5
    flat[3::7] = np.float32(2.0 ** -20)
    ref = oracle.gen_rollout(og, sent, R, u)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(ref), precision=precision).cpu().numpy()
    _cases.check_token_parity(replay, ref, u, og)
    assert replay.min() >= 0 and replay.max() < V


# ------------------------------------------------------------------ generator beyond E = H = 32 (tg_genw.hip)
WIDE_CASES = [
    # V, E, H, B, L, R, std -- the canonical sampler has more than 16 groups per span above V = 16384, B*R*L and V
    # are ragged against the 128-wide GEMM tiles, E != H
    (20011, 64, 64, 5, 7, 3, 0.3),
    (32000, 32, 128, 8, 6, 2, 0.2),
    (50257, 64, 64, 6, 5, 2, 0.2),     # 50 groups per span: the draw kernel's second instantiation
    (900, 160, 96, 7, 6, 2, 0.3),      # hidden / embedding sizes only the GEMM path takes
    (600, 32, 768, 5, 5, 2, 0.05),
    (1500, 96, 32, 40, 9, 4, 0.5),
    (4099, 512, 512, 3, 5, 2, 0.05),
]


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
@pytest.mark.parametrize("V,E,H,B,L,R,std", WIDE_CASES)
def test_rollout_wide_shapes_against_oracle(V, E, H, B, L, R, std, precision):
    """Shapes outside the E = H = 32 kernels: split_f16 runs the step-synchronous GEMM path (cell and projection
    on split-fp16 MFMAs, per-tile sampler records), fp32 the generic persistent kernel.  Step-wise token parity with
    the oracle (>= 99.9 %, disagreements only at CDF ties), prefix copies exact, sub-ranges equal to the full call."""
    gsd = synth.gen_state(V, E, H, seed=11, std=std)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    sent, u = synth.tokens(21, (B, L), V), synth.uniform01(22, (R, L, L, B))
    if precision == "fp32" and tg._lib.lib().tg_gen_supported(V, E, H) == 1:
        with pytest.raises(RuntimeError, match="not supported"):  # hidden sizes outside the fp32 kernels' set
            tg.ops.gen_rollout(gen, _t(sent), R, _t(u), precision=precision)
        return
    ref = oracle.gen_rollout(og, sent, R, u)
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(ref), precision=precision)
    prof = tg._lib.profile_read()
    tg._lib.profile_enable(False)
    wide = precision == "split_f16"  # which kernels served the call
    assert (prof["genw_proj"][1] > 0) == wide and (prof["genw_draw"][1] > 0) == wide and (prof["gen_sample"][1] > 0) != wide
    _cases.check_token_parity(replay.cpu().numpy(), ref, u, og)
    free = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), precision=precision)
    assert (free.cpu().numpy() == ref).all(-1).mean() >= 0.97
    part = torch.full_like(free, -1)
    tg.ops.gen_rollout(gen, _t(sent), R, _t(u), g_lo=2, g_hi=4, i_lo=1, i_hi=R, out=part, precision=precision)
    assert torch.equal(part[1:, 1:4], free[1:, 1:4])
    assert (part[0] == -1).all() and (part[1:, :1] == -1).all() and (part[1:, 4:] == -1).all()


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
def test_rollout_scaled_config_shape_against_oracle(precision):
    """BASELINE configs[2]'s exact generator shape (V 32 000, L 51, E = H = 512) against the oracle, at a batch the CPU
    does in seconds (B 32, rollout_num 1: 1 632 completions, 40 800 draws, 1.3 TFLOP of projection on the host):
    step-wise token parity with every disagreement a CDF tie, free-running rows, and the rewards of the rows that
    agree (SeqGAN_D scorer, eval mode) within 1e-4."""
    V, L, E, H, B, R = 32000, 51, 512, 512, 32, 1
    gsd = synth.gen_state(V, E, H, seed=0, std=0.05)
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=0)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), dsd, DEV).eval()
    og, od = oracle.Gen(gsd), oracle.Dis(dsd)
    sent, u = synth.tokens(1, (B, L), V), synth.uniform01(2, (R, L, L, B))
    o = oracle.get_reward(og, od, sent, R, u)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(o["completions"]), precision=precision)
    _cases.check_token_parity(replay.cpu().numpy(), o["completions"], u, og)
    ro = tg.ROLLOUT(gen)
    ro.gen_precision = precision
    rew, raw, comp = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), return_parts=True)
    same = (comp.cpu().numpy() == o["completions"]).all(-1).reshape(R * L, B)
    assert same.mean() >= 0.98
    np.testing.assert_allclose(raw.cpu().numpy()[same], o["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, L, R).all(-1)
    np.testing.assert_allclose(rew.cpu().numpy()[ok], o["reward"][ok], atol=1e-4, rtol=0)


@pytest.mark.parametrize("V,E,H,N,L,std", [(20011, 64, 64, 70, 6, 0.3), (700, 32, 256, 130, 5, 0.1), (900, 96, 160, 33, 5, 0.2)])
def test_sample_wide_shapes_against_oracle(V, E, H, N, L, std):
    gsd = synth.gen_state(V, E, H, seed=12, std=std)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    u = synth.uniform01(23, (L, N))
    ref = oracle.gen_sample(og, N, L, 1, u)
    for precision in GEN_PRECISIONS:
        if precision == "fp32" and tg._lib.lib().tg_gen_supported(V, E, H) == 1:
            continue  # hidden sizes outside the fp32 kernels' set: GEMM path only
        out = tg.ops.gen_sample(gen, N, L, 1, _t(u), forced=_t(ref), precision=precision).cpu().numpy()
        _cases.check_sample_token_parity(out, ref, u, og, 1)
        free = tg.ops.gen_sample(gen, N, L, 1, _t(u), precision=precision).cpu().numpy()
        assert (free == ref).all(-1).mean() >= 0.97


@pytest.mark.parametrize("V,E,H,N,L,std", [(20011, 64, 64, 70, 6, 0.3), (700, 32, 256, 130, 5, 0.1), (4099, 512, 512, 19, 4, 0.05),
                                           (900, 96, 160, 33, 5, 0.2)])
def test_nll_wide_shapes_against_oracle(V, E, H, N, L, std):
    """tg_gen_nll beyond E = H = 32: split_f16 takes the GEMM path (log-sum-exp from the tile records, the target's
    logit an fp32 dot product), fp32 the persistent kernel; both within the 1e-5 NLL bar of the oracle."""
    gsd = synth.gen_state(V, E, H, seed=14, std=std)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    tgt = synth.tokens(24, (N, L), V)
    inp = np.concatenate([np.ones((N, 1), np.int64), tgt[:, :-1]], 1)
    o = oracle.gen_teacher_forced(og, inp, tgt)
    for precision in GEN_PRECISIONS:
        if precision == "fp32" and tg._lib.lib().tg_gen_supported(V, E, H) == 1:
            continue
        tg._lib.profile_enable(True)
        tg._lib.profile_read()
        lp, lse = tg.ops.gen_nll(gen, _t(inp), _t(tgt), precision=precision, want_lse=True)
        prof = tg._lib.profile_read()
        tg._lib.profile_enable(False)
        assert (prof["genw_proj"][1] > 0) == (precision == "split_f16")
        np.testing.assert_allclose(lse.cpu().numpy(), o["lse"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(lp.cpu().numpy(), o["logp"], rtol=1e-5, atol=1e-5)
        assert abs(float(-lp.mean()) - float(-o["logp"].mean())) <= 1e-5 * abs(float(o["logp"].mean()))


@pytest.mark.parametrize("std,emb_scale,bias_scale", [(0.01, 1.0, 1.0), (2.0, 1.0, 1.0), (0.3, 1e-3, 30.0), (0.3, 50.0, 1.0)])
def test_wide_generator_across_weight_scales(std, emb_scale, bias_scale):
    """The wide path scales W_out, W_ih, W_hh and the embedding table by powers of two taken from their |max| and
    shares one scale between the two products of the cell: parity has to hold when the tensors' ranges differ."""
    V, E, H, B, L, R = 3000, 64, 128, 16, 8, 2
    gsd = {k: np.array(v) for k, v in synth.gen_state(V, E, H, seed=13, std=std).items()}
    gsd["embeddings.weight"] = (gsd["embeddings.weight"] * emb_scale).astype(np.float32)
    gsd["lstm2out.bias"] = (gsd["lstm2out.bias"] * bias_scale).astype(np.float32)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    sent, u = synth.tokens(1, (B, L), V), synth.uniform01(2, (R, L, L, B))
    ref = oracle.gen_rollout(og, sent, R, u)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(ref), precision="split_f16").cpu().numpy()
    _cases.check_token_parity(replay, ref, u, og)


# ------------------------------------------------------------------ discriminator
@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("name", _cases.names("dis"))
def test_discriminator(name, precision):
    c = _cases.load(name)
    dis = _dis(c).eval()
    seqs = synth.tokens(int(c["sseed"]), (int(c["N"]), int(c["L"])), int(c["V"]))
    r = tg.ops.dis_score(dis, _t(seqs), want_logits=True, want_feature=True, precision=precision)
    np.testing.assert_allclose(r["feature"].cpu().numpy(), c["feature"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(r["logits"].cpu().numpy(), c["logits"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(tg.discriminator.forward(dis, _t(seqs)).cpu().numpy(), c["logits"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(tg.discriminator.get_feature(dis, _t(seqs)).cpu().numpy(), c["feature"], rtol=1e-4,
                               atol=2e-6)
    if max(conv.kernel_size[0] for conv in dis.convs) > 3:
        with pytest.raises(RuntimeError):  # the reference raises from conv2d when L < filter width
            tg.ops.dis_score(dis, _t(seqs[:, :3]))


@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("N", [1, 5, 129, 300])
def test_discriminator_ragged_and_dropout(N, precision):
    V, L = 97, 23
    sd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=9)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), sd, DEV)
    od = oracle.Dis(sd)
    seqs = synth.tokens(N, (N, L), V)
    du = synth.uniform01(N + 1, (N, od.F))
    o = oracle.dis_forward(od, seqs, col=1, dropout_p=0.25, dropout_uniforms=du, want_feature=True)
    r = tg.ops.dis_score(dis, _t(seqs), col=1, dropout_p=0.25, dropout_uniforms=_t(du), want_logits=True,
                         want_feature=True, precision=precision)
    np.testing.assert_allclose(r["feature"].cpu().numpy(), o["feature"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(r["logits"].cpu().numpy(), o["logits"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(r["reward"].cpu().numpy(), o["reward"], atol=1e-5, rtol=0)


@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("Ed", [32, 48, 8])
def test_discriminator_narrower_embedding(Ed, precision):
    """CNNDiscriminator takes any embed_dim (models/discriminator.py:20-36; cfg.dis_embed_dim is a command-line flag).
    Narrower than the kernels' 64: the shim zero-pads embedding and conv-weight columns per call, which is exact;
    features, logits and rewards against the oracle run on the Ed-wide state_dict itself.  Wider: refused."""
    V, L, N = 211, 22, 45
    sd = synth.dis_state(V, Ed, *synth.SEQGAN_D_FILTERS, n_out=2, seed=12)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(Ed, V), sd, DEV)
    od = oracle.Dis(sd)
    assert od.Ed == Ed
    seqs = synth.tokens(N, (N, L), V)
    du = synth.uniform01(N + 1, (N, od.F))
    o = oracle.dis_forward(od, seqs, col=1, dropout_p=0.25, dropout_uniforms=du, want_feature=True)
    r = tg.ops.dis_score(dis, _t(seqs), col=1, dropout_p=0.25, dropout_uniforms=_t(du), want_logits=True,
                         want_feature=True, precision=precision)
    np.testing.assert_allclose(r["feature"].cpu().numpy(), o["feature"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(r["logits"].cpu().numpy(), o["logits"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(r["reward"].cpu().numpy(), o["reward"], atol=1e-5, rtol=0)
    wide = tg.modules.SeqGAN_D(96, V).to(DEV)
    with pytest.raises(RuntimeError, match="wider than"):
        tg.ops.dis_score(wide, _t(seqs))


def test_discriminator_unaligned_dropout_uniforms():
    """dropout uniforms that start 4 bytes into an allocation (a contiguous view): the 16-byte epilogue loads of the
    split highway cannot be used; the call must still match the oracle."""
    V, L, N = 97, 23, 70
    sd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=9)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), sd, DEV)
    od = oracle.Dis(sd)
    seqs = synth.tokens(N, (N, L), V)
    du = synth.uniform01(N + 1, (N, od.F))
    o = oracle.dis_forward(od, seqs, col=1, dropout_p=0.25, dropout_uniforms=du)
    buf = torch.empty(N * od.F + 1, device=DEV)
    view = buf[1:].view(N, od.F)
    view.copy_(_t(du))
    assert view.is_contiguous() and view.data_ptr() % 16 == 4
    r = tg.ops.dis_score(dis, _t(seqs), col=1, dropout_p=0.25, dropout_uniforms=view, want_logits=True)
    np.testing.assert_allclose(r["logits"].cpu().numpy(), o["logits"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(r["reward"].cpu().numpy(), o["reward"], atol=1e-5, rtol=0)


# filter tables other than the two the reference ships (CNNDiscriminator takes any, models/discriminator.py:20-36):
# F % 8 != 0 (split mode then runs the exact-fp32 highway), F % 8 == 0 with banks that start off the 4-feature grid
# (single-feature stores into the highway's B pieces), one bank, more than 8 banks' worth of remainder rows, n_out 8
@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("fs,ns,n_out,N,L", [((2, 3, 5), (33, 70, 10), 2, 77, 9),
                                             ((1, 4), (36, 68), 3, 40, 12),
                                             ((3,), (64,), 8, 300, 7),
                                             ((1, 2, 3, 4, 5, 6, 7), (5, 9, 13, 17, 21, 25, 30), 2, 33, 8),
                                             ((2, 7), (250, 6), 5, 513, 11)])
def test_discriminator_other_filter_tables(fs, ns, n_out, N, L, precision):
    V = 211
    sd = synth.dis_state(V, 64, fs, ns, n_out=n_out, seed=21)
    dis = tg.modules.load_numpy_state(tg.modules.CNNDiscriminator(64, V, list(fs), list(ns), dropout=0.3, n_out=n_out),
                                      sd, DEV)
    od = oracle.Dis(sd)
    seqs = synth.tokens(N + 3, (N, L), V)
    du = synth.uniform01(N + 4, (N, od.F))
    col = n_out - 1
    o = oracle.dis_forward(od, seqs, col=col, dropout_p=0.3, dropout_uniforms=du, want_feature=True)
    r = tg.ops.dis_score(dis, _t(seqs), col=col, dropout_p=0.3, dropout_uniforms=_t(du), want_logits=True,
                         want_feature=True, precision=precision)
    np.testing.assert_allclose(r["feature"].cpu().numpy(), o["feature"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(r["logits"].cpu().numpy(), o["logits"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(r["reward"].cpu().numpy(), o["reward"], atol=1e-5, rtol=0)


# ------------------------------------------------------------------ get_reward end to end
@pytest.mark.parametrize("name", _cases.names("get_reward"))
def test_get_reward(name):
    c = _cases.load(name)
    gen, dis = _gen(c), _dis(c)
    sent, u, du = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    if int(c["dropout"]):
        dis.train()
    else:
        dis.eval()
    ro = tg.ROLLOUT(gen, True)
    rew, raw, comp = ro.get_reward(_t(sent), R, dis, int(c["current_k"]), uniforms=_t(u),
                                   dropout_uniforms=None if du is None else _t(du), return_parts=True)
    assert rew.is_cuda and rew.dtype == torch.float32 and tuple(rew.shape) == (B, L)
    comp, raw, rew = comp.cpu().numpy(), raw.cpu().numpy(), rew.cpu().numpy()
    same = (comp == c["completions"]).all(-1).reshape(R * L, B)
    assert same.mean() >= 0.98
    np.testing.assert_allclose(raw[same], c["raw"][same], atol=1e-4, rtol=0)
    # out[b,t] = mean_r raw.flatten()[(b*L+t)*R + r] (utils/rollout.py:153): compare every output whose R contributing
    # raw entries come from completions that agree -- one cascaded CDF tie no longer hides the whole [B,L] matrix
    ok = same.reshape(B, L, R).all(-1)
    assert ok.mean() >= 0.9
    np.testing.assert_allclose(rew[ok], c["reward"][ok], atol=1e-4, rtol=0)
    # the scorer on the reference's own completions: every raw row within 1e-4, and the reduce
    d = tg.ops.dis_score(dis, _t(c["completions"].astype(np.int64).reshape(R * L * B, L)), col=int(c["current_k"]) + 1,
                         dropout_p=float(c["dropout_p"]),
                         dropout_uniforms=None if du is None else _t(du.reshape(R * L * B, -1)))
    np.testing.assert_allclose(d["reward"].cpu().numpy().reshape(R * L, B), c["raw"], atol=1e-4, rtol=0)
    red = tg.ops.reward_reduce(_t(c["raw"]), B, L, R).cpu().numpy()
    np.testing.assert_allclose(red, c["reward"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(red, c["raw"].reshape(B, L, R).mean(-1), atol=1e-6, rtol=0)


@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("name", ["reward_senti_k2", "reward_tiny_dropout", "reward_len51", "reward_c2_r1"])
def test_prefix_reuse_scorer_is_bit_identical(name, precision):
    """tg_dis_score_rollout (conv windows inside the shared prefix evaluated once per sentence) against the plain
    scorer on the same completions: identical bits, including B not a multiple of the workgroup's sequence count."""
    c = _cases.load(name)
    dis = _dis(c)
    sent, u, du = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    comp = _t(c["completions"].astype(np.int64))
    p = float(c["dropout_p"])
    dut = None if du is None else _t(du.reshape(R * L * B, -1))
    plain = tg.ops.dis_score(dis, comp.view(R * L * B, L), col=int(c["current_k"]) + 1, dropout_p=p,
                             dropout_uniforms=dut, precision=precision)["reward"].view(R * L, B)
    fast = tg.ops.dis_score_rollout(dis, _t(sent), comp, col=int(c["current_k"]) + 1, dropout_p=p,
                                    dropout_uniforms=dut, precision=precision)
    assert torch.equal(plain, fast)
    np.testing.assert_allclose(fast.cpu().numpy(), c["raw"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
def test_scorer_preparation_beside_the_generator_gives_the_same_bits(precision):
    """ROLLOUT.get_reward enqueues the scorer's preparation (tg_dis_score_rollout_prepare: weight pieces, the sentences'
    window values) on a second stream beside the generator and finishes with tg_dis_score_rollout_prepared: the rewards
    are those of the one-call path to the bit, with live dropout from explicit uniforms, call after call (the side
    stream's workspace is handed back and reused), and on a misaligned dropout tensor (exact-fp32 highway)."""
    c = _cases.load("reward_c1_r2")
    gen, dis = _gen(c), _dis(c)
    dis.train()
    sent, u, _ = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    F = int(tg._lib.dis_weights(dis, precision)[0].F)
    du = _t(synth.uniform01(17, (R * L * B * F + 1,)))
    ro = tg.ROLLOUT(gen)
    ro.dis_precision = precision
    for dropout_uniforms in (du[:-1].view(R * L, B, F), du[1:].view(R * L, B, F)):   # 16-byte aligned / off by 4 bytes
        ro.overlap_prepare = False
        want = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), dropout_uniforms=dropout_uniforms, return_parts=True)
        ro.overlap_prepare = True
        for _ in range(3):
            got = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), dropout_uniforms=dropout_uniforms, return_parts=True)
            assert torch.equal(got[1], want[1]) and torch.equal(got[0], want[0]) and torch.equal(got[2], want[2])
    # the two halves called by hand, with the preparation on the current stream
    w = tg._lib.dis_weights(dis, precision)
    ws = tg.ops.dis_score_rollout_prepare(dis, _t(sent), R, weights=w)
    raw = tg.ops.dis_score_rollout(dis, _t(sent), want[2], col=1, dropout_p=float(dis.dropout.p),
                                   dropout_uniforms=du[:-1].view(R * L * B, F), weights=w, prepared=ws)
    ro.overlap_prepare = False
    assert torch.equal(raw, ro.get_reward(_t(sent), R, dis, uniforms=_t(u), dropout_uniforms=du[:-1].view(R * L, B, F),
                                          return_parts=True)[1])


def test_split_precision_against_fp32():
    """The default split-fp16 x3 scorer against the exact-fp32 scorer on the same inputs at C2's shape: pooled
    features / logits within 2e-5 relative (measured ~2e-6), rewards within 1e-5."""
    V, L, N = 5255, 51, 2048
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=3, device=DEV).eval()
    seqs = _t(synth.tokens(5, (N, L), V))
    a = tg.ops.dis_score(dis, seqs, want_logits=True, want_feature=True, precision="fp32")
    b = tg.ops.dis_score(dis, seqs, want_logits=True, want_feature=True, precision="split_f16")
    scale = float(a["feature"].abs().max())
    assert float((a["feature"] - b["feature"]).abs().max()) <= 2e-5 * scale
    assert float((a["logits"] - b["logits"]).abs().max()) <= 2e-5 * max(1.0, float(a["logits"].abs().max()))
    assert float((a["reward"] - b["reward"]).abs().max()) <= 1e-5


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
def test_get_reward_c1_full_vs_oracle(precision):
    """BASELINE config 1 at full size (V5000 L20 E=H=32 B64 R16): 20 480 completions against the
    oracle; step-wise token parity, and rewards on rows whose completions agree."""
    V, L, B, R = 5000, 20, 64, 16
    gsd = synth.gen_state(V, 32, 32, seed=0, std=1.0)
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=0)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), dsd, DEV).eval()
    sent = synth.tokens(1, (B, L), V)
    u = synth.uniform01(2, (R, L, L, B))
    og, od = oracle.Gen(gsd), oracle.Dis(dsd)
    o = oracle.get_reward(og, od, sent, R, u)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(o["completions"]), precision=precision).cpu().numpy()
    _cases.check_token_parity(replay, o["completions"], u, og)
    ro = tg.ROLLOUT(gen)
    ro.gen_precision = precision
    rew, raw, comp = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), return_parts=True)
    same = (comp.cpu().numpy() == o["completions"]).all(-1).reshape(R * L, B)
    assert same.mean() >= 0.98
    np.testing.assert_allclose(raw.cpu().numpy()[same], o["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, L, R).all(-1)
    assert ok.mean() >= 0.9
    np.testing.assert_allclose(rew.cpu().numpy()[ok], o["reward"][ok], atol=1e-4, rtol=0)
    # size-independent properties: rewards are probabilities, the g = L rows do not depend on i
    rw = raw.cpu().numpy().reshape(R, L, B)
    assert np.all((rw >= 0) & (rw <= 1))
    assert np.array_equal(rw[:, L - 1], np.broadcast_to(rw[0, L - 1], (R, B)))
    assert abs(float(rew.mean()) - float(o["reward"].mean())) < 1e-3


_C2_ORACLE = {}   # the oracle's full-size configs[1] call, computed once for both precisions


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
def test_get_reward_c2_full_vs_oracle(precision):
    """BASELINE configs[1] -- the headline workload -- at FULL size (V5255 L51 E=H=32 B64 rollout_num 16: 52 224
    completions, 1 305 600 draws, live dropout p = 0.25 from explicit uniforms) against the oracle (about 7 s on the
    box's 16 host cores): step-wise token parity with every disagreement a CDF tie, raw rows of agreeing completions
    and the [B,L] rewards they feed within 1e-4, in both generator precisions."""
    V, L, B, R = 5255, 51, 64, 16
    gsd = synth.gen_state(V, 32, 32, seed=0, std=1.0)
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=0)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), dsd, DEV).train()
    sent = synth.tokens(1, (B, L), V)
    u = synth.uniform01(2, (R, L, L, B))
    og, od = oracle.Gen(gsd), oracle.Dis(dsd)
    # dropout uniforms drawn on the device and handed to both sides (R*L*B*F = 90 M values)
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    du_t = torch.rand(R * L, B, od.F, device=DEV, generator=g)
    if "o" not in _C2_ORACLE:
        _C2_ORACLE["o"] = oracle.get_reward(og, od, sent, R, u, dropout_p=0.25, dropout_uniforms=du_t.cpu().numpy())
    o = _C2_ORACLE["o"]
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(o["completions"]), precision=precision).cpu().numpy()
    _cases.check_token_parity(replay, o["completions"], u, og)
    ro = tg.ROLLOUT(gen)
    ro.gen_precision = precision
    rew, raw, comp = ro.get_reward(_t(sent), R, dis, uniforms=_t(u), dropout_uniforms=du_t, return_parts=True)
    same = (comp.cpu().numpy() == o["completions"]).all(-1).reshape(R * L, B)
    assert same.mean() >= 0.98
    np.testing.assert_allclose(raw.cpu().numpy()[same], o["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, L, R).all(-1)
    assert ok.mean() >= 0.75          # an output averages 16 completions: a 1 % row disagreement touches ~15 % of them
    np.testing.assert_allclose(rew.cpu().numpy()[ok], o["reward"][ok], atol=1e-4, rtol=0)
    assert abs(float(rew.mean()) - float(o["reward"].mean())) < 1e-3


def test_get_reward_c4_full_vs_oracle():
    """BASELINE config 4 at full size for one of its k = 4 generators (V5000 L20 E=H=32 B64 R16, SentiGAN's 5-way
    scorer F = 800 with live dropout p = 0.2, column current_k + 1 = 3): 20 480 completions against the oracle."""
    V, L, B, R, cur_k = 5000, 20, 64, 16, 2
    gsd = synth.gen_state(V, 32, 32, seed=cur_k, std=1.0)
    dsd = synth.dis_state(V, 64, *synth.SENTIGAN_D_FILTERS, n_out=5, seed=0)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SentiGAN_D(4, 64, V), dsd, DEV)
    assert dis.training and dis.dropout.p == 0.2
    sent = synth.tokens(1, (B, L), V)
    u = synth.uniform01(2, (R, L, L, B))
    og, od = oracle.Gen(gsd), oracle.Dis(dsd)
    du = synth.uniform01(3, (R * L, B, od.F))
    o = oracle.get_reward(og, od, sent, R, u, col=cur_k + 1, dropout_p=0.2, dropout_uniforms=du)
    rew, raw, comp = tg.ROLLOUT(gen).get_reward(_t(sent), R, dis, cur_k, uniforms=_t(u), dropout_uniforms=_t(du),
                                                return_parts=True)
    same = (comp.cpu().numpy() == o["completions"]).all(-1).reshape(R * L, B)
    assert same.mean() >= 0.98
    np.testing.assert_allclose(raw.cpu().numpy()[same], o["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, L, R).all(-1)
    assert ok.mean() >= 0.9
    np.testing.assert_allclose(rew.cpu().numpy()[ok], o["reward"][ok], atol=1e-4, rtol=0)
    assert np.all((raw.cpu().numpy() >= 0) & (raw.cpu().numpy() <= 1))


@pytest.mark.parametrize("std,bias_scale", [(0.01, 1.0), (4.0, 1.0), (1.0, 50.0), (1e-4, 1e-3)])
def test_generator_split_precision_across_weight_scales(std, bias_scale):
    """The split projection scales its operands by a power of two taken from |max|: token parity with the oracle has
    to hold for tiny, large and bias-dominated output layers alike."""
    V, L, B, R = 5000, 20, 64, 2
    gsd = {k: np.array(v) for k, v in synth.gen_state(V, 32, 32, seed=3, std=std).items()}
    gsd["lstm2out.bias"] = (gsd["lstm2out.bias"] * bias_scale).astype(np.float32)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    sent, u = synth.tokens(1, (B, L), V), synth.uniform01(2, (R, L, L, B))
    og = oracle.Gen(gsd)
    ref = oracle.gen_rollout(og, sent, R, u)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(ref), precision="split_f16").cpu().numpy()
    _cases.check_token_parity(replay, ref, u, og)


@pytest.mark.parametrize("conv_scale,emb_scale", [(1e-3, 1.0), (1e3, 1.0), (1.0, 1e-3), (30.0, 30.0)])
def test_scorer_split_precision_across_weight_scales(conv_scale, emb_scale):
    V, L, N = 5000, 20, 512
    sd = {k: np.array(v) for k, v in synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=5).items()}
    for k in sd:
        if k.startswith("convs") and k.endswith("weight"):
            sd[k] = (sd[k] * conv_scale).astype(np.float32)
    sd["embeddings.weight"] = (sd["embeddings.weight"] * emb_scale).astype(np.float32)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), sd, DEV).eval()
    seqs = synth.tokens(7, (N, L), V)
    a = tg.ops.dis_score(dis, _t(seqs), want_feature=True, precision="fp32")
    b = tg.ops.dis_score(dis, _t(seqs), want_feature=True, precision="split_f16")
    o = oracle.dis_forward(oracle.Dis(sd), seqs, col=1, want_feature=True)
    scale = float(a["feature"].abs().max())
    assert float((a["feature"] - b["feature"]).abs().max()) <= 3e-5 * scale
    assert np.abs(b["feature"].cpu().numpy() - o["feature"]).max() <= 3e-5 * scale
    np.testing.assert_allclose(b["reward"].cpu().numpy(), o["reward"], atol=1e-4, rtol=0)


def _hash32(x):
    x = x.astype(np.uint64)
    m = np.uint64(0xffffffff)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & m
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & m
    x ^= x >> np.uint64(16)
    return x


def test_counter_based_dropout_masks():
    """dropout_uniforms = None: the masks come from tg_hash32(n*F + f + seed*0x9E3779B9) < (1-p)*2^32 inside the
    highway epilogue.  The same masks rebuilt in numpy and fed to the oracle as 0/1 "uniforms" must give the same
    logits; the keep rate is 1-p; a seed repeats, another seed differs; the MaliGAN re-scoring indexes (r, n, f)."""
    V, L, N, p, seed = 300, 20, 333, 0.25, 0xC0FFEE
    sd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=9)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), sd, DEV)
    od = oracle.Dis(sd)
    F = od.F
    seqs = synth.tokens(3, (N, L), V)
    idx = (np.arange(N, dtype=np.uint64)[:, None] * np.uint64(F) + np.arange(F, dtype=np.uint64)[None, :])
    keep = _hash32((idx + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xffffffff)) < np.uint64(int((1 - p) * 2 ** 32))
    assert abs(keep.mean() - (1 - p)) < 5e-3
    for precision in ("fp32", "split_f16"):
        r = tg.ops.dis_score(dis, _t(seqs), col=1, dropout_p=p, want_logits=True, dropout_seed=seed, precision=precision)
        o = oracle.dis_forward(od, seqs, col=1, dropout_p=p, dropout_uniforms=np.where(keep, 0.0, 1.0).astype(np.float32))
        np.testing.assert_allclose(r["logits"].cpu().numpy(), o["logits"], rtol=1e-4, atol=5e-6)
        again = tg.ops.dis_score(dis, _t(seqs), col=1, dropout_p=p, dropout_seed=seed, precision=precision)["reward"]
        other = tg.ops.dis_score(dis, _t(seqs), col=1, dropout_p=p, dropout_seed=seed + 1, precision=precision)["reward"]
        assert torch.equal(again, r["reward"]) and not torch.equal(other, r["reward"])
    R = 3
    feat = tg.ops.dis_score(dis, _t(seqs), want_reward=False, want_feature=True, precision="fp32")["feature"]
    probs = tg.ops.dis_rescore_dropout(dis, feat, R, col=1, dropout_p=p, dropout_seed=seed).cpu().numpy()
    idx3 = (np.arange(R * N, dtype=np.uint64)[:, None] * np.uint64(F) + np.arange(F, dtype=np.uint64)[None, :])
    keep3 = _hash32((idx3 + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xffffffff)) < np.uint64(int((1 - p) * 2 ** 32))
    u3 = np.where(keep3, 0.0, 1.0).astype(np.float32).reshape(R, N, F)
    ref = tg.ops.dis_rescore_dropout(dis, feat, R, col=1, dropout_p=p, dropout_uniforms=_t(u3)).cpu().numpy()
    np.testing.assert_array_equal(probs, ref)
    # the drop-in draws a fresh seed per call: two get_reward calls with live dropout differ, eval mode does not
    gen = tg.modules.synthetic_generator(V, 32, 32, L, seed=1, std=0.3, device=DEV)
    sent = _t(synth.tokens(4, (8, L), V))
    u = torch.rand(2, L, L, 8, device=DEV)
    ro = tg.ROLLOUT(gen)
    a1 = ro.get_reward(sent, 2, dis.train(), uniforms=u)
    a2 = ro.get_reward(sent, 2, dis, uniforms=u)
    assert not torch.equal(a1, a2) and float((a1 - a2).abs().max()) < 0.5
    e1 = ro.get_reward(sent, 2, dis.eval(), uniforms=u)
    assert torch.equal(e1, ro.get_reward(sent, 2, dis, uniforms=u))


def test_get_reward_c2_full_size_properties():
    """BASELINE config 2 at full size (V5255 L51 E=H=32 B64 R16, 52 224 completions, live dropout): properties that
    do not need the oracle.  Determinism, the sentence rows, probabilities, sharding by rollout index (each shard
    computed on its own equals the unsharded call bit for bit), prefix copy, and the reduce's closed form."""
    V, L, B, R = 5255, 51, 64, 16
    gen = tg.modules.synthetic_generator(V, 32, 32, L, seed=0, std=1.0, device=DEV)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=DEV).train()
    sent = _t(synth.tokens(1, (B, L), V))
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    u = torch.rand(R, L, L, B, device=DEV, generator=g)
    du = torch.rand(R * L, B, dis.feature_dim, device=DEV, generator=g)
    ro = tg.ROLLOUT(gen)
    rew, raw, comp = ro.get_reward(sent, R, dis, uniforms=u, dropout_uniforms=du, return_parts=True)
    rew2, raw2, comp2 = ro.get_reward(sent, R, dis, uniforms=u, dropout_uniforms=du, return_parts=True)
    assert torch.equal(comp, comp2) and torch.equal(raw, raw2) and torch.equal(rew, rew2)
    assert tuple(comp.shape) == (R, L, B, L) and int(comp.min()) >= 0 and int(comp.max()) < V
    for gnum in (1, 17, L):  # the first given_num tokens are the sentence's
        assert torch.equal(comp[:, gnum - 1, :, :gnum], sent[:, :gnum].expand(R, B, gnum))
    assert bool(((raw >= 0) & (raw <= 1)).all())
    np.testing.assert_allclose(rew.cpu().numpy(), raw.cpu().numpy().reshape(B, L, R).mean(-1), atol=1e-6, rtol=0)
    # shards of the rollout index, each on its own, against the unsharded completions / raw rows
    for lo, hi in ((0, 4), (4, 5), (5, 16)):
        part = torch.full_like(comp, -1)
        tg.ops.gen_rollout(gen, sent, R, u, i_lo=lo, i_hi=hi, out=part)
        assert torch.equal(part[lo:hi], comp[lo:hi]) and bool((part[:lo] == -1).all()) and bool((part[hi:] == -1).all())
        dshard = du.view(R, L * B * dis.feature_dim)[lo:hi].reshape((hi - lo) * L * B, dis.feature_dim)
        r = tg.ops.dis_score_rollout(dis, sent, comp[lo:hi].contiguous(), col=1, dropout_p=float(dis.dropout.p),
                                     dropout_uniforms=dshard.contiguous())
        assert torch.equal(r, raw[lo * L:hi * L])
    # eval mode: the g = L rows score the sentence itself and cannot depend on the rollout index
    dis.eval()
    _, raw_e, _ = ro.get_reward(sent, R, dis, uniforms=u, return_parts=True)
    rw = raw_e.view(R, L, B)
    assert torch.equal(rw[:, L - 1], rw[0, L - 1].expand(R, B))


def test_wide_rollout_across_chunks_against_oracle():
    """The GEMM path advances at most 2^18 completions together: B*L*R = 409 600 here, two chunks of rollout indices
    (25 + 15).  Token parity with the oracle over the whole call, and the second chunk on its own equals the full call."""
    V, E, H, B, L, R = 700, 32, 64, 512, 20, 40
    gsd = synth.gen_state(V, E, H, seed=15, std=0.3)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    sent, u = synth.tokens(25, (B, L), V), synth.uniform01(26, (R, L, L, B))
    ref = oracle.gen_rollout(og, sent, R, u)
    replay = tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(ref), precision="split_f16")
    _cases.check_token_parity(replay.cpu().numpy(), ref, u, og)
    part = torch.full_like(replay, -1)
    tg.ops.gen_rollout(gen, _t(sent), R, _t(u), forced=_t(ref), i_lo=25, i_hi=R, out=part, precision="split_f16")
    assert torch.equal(part[25:], replay[25:]) and bool((part[:25] == -1).all())


def test_get_reward_scaled_shape_properties():
    """BASELINE config 3's shape (V32000 L51 E=H=512 B4096) at rollout_num 2 -- two chunks of the GEMM path, 417 792
    completions: determinism, prefix copies, shards by rollout index equal to the unsharded call bit for bit, rewards
    are probabilities, the given_num = L rows do not depend on the rollout index."""
    V, L, E, H, B, R = 32000, 51, 512, 512, 4096, 2
    gen = tg.modules.synthetic_generator(V, E, H, L, seed=0, std=0.05, device=DEV)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=DEV).eval()
    sent = _t(synth.tokens(1, (B, L), V))
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    u = torch.rand(R, L, L, B, device=DEV, generator=g)
    ro = tg.ROLLOUT(gen)
    rew, raw, comp = ro.get_reward(sent, R, dis, uniforms=u, return_parts=True)
    rew2, raw2, comp2 = ro.get_reward(sent, R, dis, uniforms=u, return_parts=True)
    assert torch.equal(comp, comp2) and torch.equal(raw, raw2) and torch.equal(rew, rew2)
    assert tuple(comp.shape) == (R, L, B, L) and int(comp.min()) >= 0 and int(comp.max()) < V
    for gnum in (1, 23, L):
        assert torch.equal(comp[:, gnum - 1, :, :gnum], sent[:, :gnum].expand(R, B, gnum))
    assert not torch.equal(comp[0, 0], comp[1, 0])  # different uniforms, different completions
    assert bool(((raw >= 0) & (raw <= 1)).all())
    rw = raw.view(R, L, B)
    assert torch.equal(rw[:, L - 1], rw[0, L - 1].expand(R, B))
    part = torch.full_like(comp, -1)
    tg.ops.gen_rollout(gen, sent, R, u, i_lo=1, i_hi=2, out=part)
    assert torch.equal(part[1], comp[1]) and bool((part[0] == -1).all())
    # every sampled token is a draw from the whole vocabulary: over 5 M draws of a near-flat distribution the
    # top and bottom halves of the id range both occur in quantity
    drawn = comp[0, 0, :, 1:]
    assert 0.4 < float((drawn < V // 2).float().mean()) < 0.6


@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("name", _cases.names("mali_reward"))
def test_mali_reward(name, precision):
    """tg.maligan.get_mali_reward (features once + tg_dis_rescore_dropout) against the reference's
    get_mali_reward golden and the oracle; values are O(1/B), the tolerance is relative to that scale."""
    c = _cases.load(name)
    V, L, B, R = int(c["V"]), int(c["L"]), int(c["B"]), int(c["R"])
    sd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=int(c["dseed"]))
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V), sd, DEV).train()
    samples = synth.tokens(int(c["sseed"]), (B, L), V)
    du = synth.uniform01(int(c["useed"]), (R, B, dis.feature_dim))
    rew = tg.maligan.get_mali_reward(dis, _t(samples), R, dropout_uniforms=_t(du), precision=precision)
    assert rew.is_cuda and tuple(rew.shape) == (B, L)
    rew = rew.cpu().numpy()
    np.testing.assert_allclose(rew, c["reward"], atol=1e-4 / B, rtol=1e-4)
    np.testing.assert_allclose(rew, oracle.mali_reward(oracle.Dis(sd), samples, R, float(c["dropout_p"]), du),
                               atol=1e-4 / B, rtol=1e-4)
    # per-mask probabilities against R separate scorer passes with the same masks (the reference's loop)
    feat = tg.ops.dis_score(dis, _t(samples), want_reward=False, want_feature=True, precision=precision)["feature"]
    probs = tg.ops.dis_rescore_dropout(dis, feat, R, col=1, dropout_p=float(c["dropout_p"]), dropout_uniforms=_t(du))
    for r in range(R):
        one = tg.ops.dis_score(dis, _t(samples), col=1, dropout_p=float(c["dropout_p"]), dropout_uniforms=_t(du[r]),
                               precision=precision)["reward"]
        np.testing.assert_allclose(probs[r].cpu().numpy(), one.cpu().numpy(), atol=2e-6, rtol=0)
    # eval mode: no dropout, every mask row equal
    dis.eval()
    ev = tg.maligan.get_mali_reward(dis, _t(samples), R)
    assert torch.isfinite(ev).all() and abs(float(ev[:, 0].sum())) < 1e-6


# ------------------------------------------------------------------ LeakGAN rollout
def _leak_models(c):
    gsd, dsd, sent, u, du = _cases.leak_inputs(c)
    V, E, H, K, L, B, step = [int(c[k]) for k in ("V", "E", "H", "K", "L", "B", "step")]
    gen = tg.modules.load_numpy_state(tg.modules.LeakGAN_G(E, H, V, L, goal_size=K, step_size=step, batch_size=B), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V, dropout=0.2), dsd, DEV)
    return gen, dis, oracle.Leak(gsd, step), oracle.Dis(dsd), sent, u, du


@pytest.mark.parametrize("precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("name", _cases.names("leak_reward"))
def test_leak_rollout_and_reward(name, precision):
    """tg_leak_rollout / ROLLOUT.get_reward_leakgan against the reference's rollout_mc_search_leakgan /
    get_reward_leakgan goldens (tools/make_goldens_leak.py) and the oracle; `precision` is the scorer's arithmetic
    (the manager's input and the final rewards), the generator side is fp32 / fp64."""
    c = _cases.load(name)
    gen, dis, og, od, sent, u, du = _leak_models(c)
    R, L, B, step = int(c["R"]), int(c["L"]), int(c["B"]), int(c["step"])
    T = L // step
    given = [t * step + 1 for t in range(T)]
    ref = c["completions"].astype(np.int64)
    replay = tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), forced=_t(ref), dis_precision=precision)
    _cases.check_leak_token_parity(replay.cpu().numpy(), ref, u, og, od)
    free = tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), dis_precision=precision)
    assert (free.cpu().numpy() == ref).all(-1).mean() >= 0.97
    if R > 1:  # a shard of the rollout index on its own
        part = torch.full_like(free, -1)
        tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), i_lo=1, i_hi=R, out=part, dis_precision=precision)
        assert torch.equal(part[1:], free[1:]) and bool((part[0] == -1).all())
    # the drop-in surface
    if int(c["dropout"]):
        dis.train()
    else:
        dis.eval()
    ro = tg.ROLLOUT(gen, True)
    ro.dis_precision = precision
    assert ro.step_size == step and ro.goal_out_size == od.F
    rew, raw, comp = ro.get_reward_leakgan(_t(sent), R, dis, int(c["current_k"]), uniforms=_t(u),
                                           dropout_uniforms=None if du is None else _t(du), return_parts=True)
    assert rew.is_cuda and tuple(rew.shape) == (B, T) and torch.equal(comp, free)
    # the final scoring's preparation runs on a second stream beside the rollout (tg_dis_score_prepare / _prepared):
    # the same bits as the one-call path
    ro.overlap_prepare = False
    rew1, raw1, _ = ro.get_reward_leakgan(_t(sent), R, dis, int(c["current_k"]), uniforms=_t(u),
                                          dropout_uniforms=None if du is None else _t(du), return_parts=True)
    ro.overlap_prepare = True
    if du is not None or not int(c["dropout"]):   # (counter-based masks take a fresh seed per call)
        assert torch.equal(raw1, raw) and torch.equal(rew1, rew)
    comp, raw, rew = comp.cpu().numpy(), raw.cpu().numpy(), rew.cpu().numpy()
    same = (comp == ref).all(-1).reshape(R * T, B)
    np.testing.assert_allclose(raw[same], c["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, T, R).all(-1)
    assert ok.mean() >= 0.9
    np.testing.assert_allclose(rew[ok], c["reward"][ok], atol=1e-4, rtol=0)
    one = ro.rollout_mc_search_leakgan(_t(sent), dis, given[1], uniforms=_t(u[:1, 1:2]))
    assert torch.equal(one, free[0, 1])


@pytest.mark.parametrize("gen_precision", ["fp32", "split_f16"])
def test_leak_reward_c1_shape_full_vs_oracle(gen_precision):
    """run_leakgan.py's configuration at full size (V5000 L20 E=H=32 goal 16 step 4, B64, rollout_num 4: 1 280
    completions, 14 080 draws) against the oracle: step-wise token parity with CDF-tie check, rewards of agreeing rows.
    gen_precision: the work2goal projection exact in fp32 or as split-fp16 products (TgLeakWeights.precision)."""
    V, E, H, K, L, B, R, step = 5000, 32, 32, 16, 20, 64, 4, 4
    gsd = synth.leak_state(V, E, H, K, 1720, B, seed=7, std=0.2)
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=7)
    gen = tg.modules.load_numpy_state(tg.modules.LeakGAN_G(E, H, V, L, goal_size=K, step_size=step, batch_size=B), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V, dropout=0.2), dsd, DEV).eval()
    og, od = oracle.Leak(gsd, step), oracle.Dis(dsd)
    T = L // step
    sent, u = synth.tokens(71, (B, L), V), synth.uniform01(72, (R, T, L, B))
    o = oracle.get_reward_leakgan(og, od, sent, R, u)
    given = [t * step + 1 for t in range(T)]
    replay = tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), forced=_t(o["completions"]), precision=gen_precision)
    _cases.check_leak_token_parity(replay.cpu().numpy(), o["completions"], u, og, od)
    ro = tg.ROLLOUT(gen)
    ro.gen_precision = gen_precision
    rew, raw, comp = ro.get_reward_leakgan(_t(sent), R, dis, 0, uniforms=_t(u), return_parts=True)
    same = (comp.cpu().numpy() == o["completions"]).all(-1).reshape(R * T, B)
    assert same.mean() >= 0.97
    np.testing.assert_allclose(raw.cpu().numpy()[same], o["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, T, R).all(-1)
    assert ok.mean() >= 0.9
    np.testing.assert_allclose(rew.cpu().numpy()[ok], o["reward"][ok], atol=1e-4, rtol=0)
    for t in range(T):  # the first given_num tokens are the sentence's
        assert torch.equal(comp[:, t, :, :given[t]], _t(sent)[:, :given[t]].expand(R, B, given[t]))


@pytest.mark.parametrize("gen_precision", ["fp32", "split_f16"])
@pytest.mark.parametrize("H", [32, 64])
def test_leak_rollout_given_nums_in_any_order_and_stream_modes(H, gen_precision, monkeypatch):
    """tg_leak_rollout shares the sentence-fed part of every row with one leader row per sentence, orders its rows by
    given_num and spreads the call over four streams.  None of that may show in the tokens: a call with given_nums
    unsorted, repeated and equal to L gives, for each of them, exactly the rows the plain ascending call gives for that
    given_num under the same uniforms (the sentence itself for given_num = L); and the call with everything on one
    stream (TG_LEAK_ONE_STREAM=1) gives the same bits.  H = 32 takes the worker cell inside the draw kernel, H = 64 the
    kernel of its own; TG_LEAK_SPLIT_CELL=1 forces the latter (different summation order: rows agree up to CDF ties).
    A call cut into several chunks of the rollout index (each with its own leaders) gives the same bits too."""
    V, E, K, L, B, R, step = 300, 32, 16, 20, 7, 3, 4
    gsd = synth.leak_state(V, E, H, K, 1720, B, seed=5, std=0.2)
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=5)
    gen = tg.modules.load_numpy_state(tg.modules.LeakGAN_G(E, H, V, L, goal_size=K, step_size=step, batch_size=B), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V, dropout=0.2), dsd, DEV).eval()
    T = L // step
    given = [t * step + 1 for t in range(T)]
    sent, u = synth.tokens(91, (B, L), V), synth.uniform01(92, (R, T, L, B))
    std = tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), precision=gen_precision)
    mixed = [9, 1, L, 9, 17, 5, 13]
    um = np.stack([u[:, given.index(g)] if g in given else u[:, 0] for g in mixed], axis=1)
    got = tg.ops.leak_rollout(gen, dis, _t(sent), R, mixed, _t(np.ascontiguousarray(um)), precision=gen_precision)
    for j, g in enumerate(mixed):
        if g == L:
            assert torch.equal(got[:, j], _t(sent).expand(R, B, L))
        else:
            assert torch.equal(got[:, j], std[:, given.index(g)]), "given_num %d at index %d" % (g, j)
    # a shard of the rollout index: its own leaders, the same rows
    part = torch.full_like(std, -1)
    tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), i_lo=2, i_hi=3, out=part, precision=gen_precision)
    assert torch.equal(part[2], std[2]) and bool((part[:2] == -1).all())
    monkeypatch.setenv("TG_LEAK_ONE_STREAM", "1")
    assert torch.equal(tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), precision=gen_precision), std)
    monkeypatch.delenv("TG_LEAK_ONE_STREAM")
    # several chunks of the rollout index in one call (TG_LEAK_LOGITS_BYTES: room for one rollout index per chunk)
    monkeypatch.setenv("TG_LEAK_LOGITS_BYTES", str((B + T * B) * V * 4))
    assert torch.equal(tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), precision=gen_precision), std)
    monkeypatch.delenv("TG_LEAK_LOGITS_BYTES")
    monkeypatch.setenv("TG_LEAK_SPLIT_CELL", "1")
    other = tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), precision=gen_precision)
    assert (other == std).all(-1).float().mean() >= 0.97
    if H > 32:
        assert torch.equal(other, std)   # the switch changes nothing where the cell has its own kernel anyway


@pytest.mark.parametrize("V,E,H,K,L,B,R,step", [(3, 32, 32, 16, 20, 1, 1, 4), (40, 32, 64, 8, 20, 3, 2, 5),
                                                (150, 64, 128, 16, 24, 33, 1, 3), (9, 32, 32, 1, 20, 2, 3, 10),
                                                (64, 48, 96, 5, 21, 5, 2, 7), (40009, 32, 32, 4, 20, 2, 1, 10)])
def test_leak_reward_edge_shapes_against_oracle(V, E, H, K, L, B, R, step):
    """get_reward_leakgan on edge shapes: one sentence / one rollout, a three-word vocabulary, goal_size 1 and 5, hidden
    sizes up to the supported 128, embedding width off the 32 grid, step sizes that do not divide L (the reference then
    scores L // step_size prefixes), batches off every tile size, a vocabulary above 32 768 (the sampler's tile sums no
    longer need the row's weights in LDS) -- against the oracle, live dropout included."""
    gsd = synth.leak_state(V, E, H, K, 1720, B, seed=V + L, std=0.2)
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=3)
    gen = tg.modules.load_numpy_state(tg.modules.LeakGAN_G(E, H, V, L, goal_size=K, step_size=step, batch_size=B), gsd, DEV)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V, dropout=0.2), dsd, DEV)
    og, od = oracle.Leak(gsd, step), oracle.Dis(dsd)
    T = L // step
    sent, u = synth.tokens(81, (B, L), V), synth.uniform01(82, (R, T, L, B))
    du = synth.uniform01(83, (R * T, B, od.F))
    o = oracle.get_reward_leakgan(og, od, sent, R, u, dropout_p=0.2, dropout_uniforms=du)
    given = [t * step + 1 for t in range(T)]
    for gp in ("fp32", "split_f16"):   # the work2goal projection in both arithmetic modes (K*H + K off every tile size)
        replay = tg.ops.leak_rollout(gen, dis, _t(sent), R, given, _t(u), forced=_t(o["completions"]), precision=gp)
        _cases.check_leak_token_parity(replay.cpu().numpy(), o["completions"], u, og, od)
    rew, raw, comp = tg.ROLLOUT(gen).get_reward_leakgan(_t(sent), R, dis, 0, uniforms=_t(u), dropout_uniforms=_t(du),
                                                        return_parts=True)
    assert tuple(rew.shape) == (B, T)
    same = (comp.cpu().numpy() == o["completions"]).all(-1).reshape(R * T, B)
    np.testing.assert_allclose(raw.cpu().numpy()[same], o["raw"][same], atol=1e-4, rtol=0)
    ok = same.reshape(B, T, R).all(-1)
    assert ok.mean() >= 0.9
    np.testing.assert_allclose(rew.cpu().numpy()[ok], o["reward"][ok], atol=1e-4, rtol=0)


@pytest.mark.parametrize("ns,batch", [(1, 1), (1, 64), (5, 64), (64, 64), (130, 64), (7, 3)])
def test_sample_num_batch_rule_and_edges(ns, batch):
    """LSTMGenerator.sample's batching rule (generator.py:67: num_samples // batch_size + 1 batches unless the two are
    equal, then truncate) on its edges, tokens against the oracle."""
    V, L = 77, 20
    gsd = synth.gen_state(V, 32, 32, seed=6, std=1.0)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    num_batch = ns // batch + 1 if ns != batch else 1
    N = num_batch * batch
    u = synth.uniform01(61, (L, N))
    og = oracle.Gen(gsd)
    ref_all = oracle.gen_sample(og, N, L, 1, u)
    ref = ref_all[:ns]
    out = tg.generator.sample(gen, ns, batch, 1, uniforms=_t(u))
    assert out.device.type == "cpu" and out.dtype == torch.int64 and tuple(out.shape) == (ns, L)
    assert (out.numpy() == ref).all(-1).mean() >= 0.97 and (out.numpy() == ref).mean() >= 0.99
    # step-wise under the oracle's own tokens, both arithmetic modes: every disagreement must be a CDF tie
    for precision in GEN_PRECISIONS:
        replay = tg.ops.gen_sample(gen, N, L, 1, _t(u), forced=_t(ref_all), precision=precision).cpu().numpy()
        _cases.check_sample_token_parity(replay, ref_all, u, og, 1)


def test_nll_edge_loaders():
    """cal_nll over loaders the reference's GenDataIter can produce at its edges: a single batch of one sequence, batches
    of unequal size (the mean is over batches, not tokens: metrics/nll.py:58-62), a one-token vocabulary tail."""
    V, L = 45, 20
    gsd = synth.gen_state(V, 32, 32, seed=8, std=0.7)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    for sizes in ([1], [3, 1], [64, 64, 5]):
        batches, want = [], []
        for i, n in enumerate(sizes):
            tgt = synth.tokens(70 + i, (n, L), V)
            inp = np.concatenate([np.ones((n, 1), np.int64), tgt[:, :-1]], 1)
            batches.append({"input": torch.from_numpy(inp), "target": torch.from_numpy(tgt)})
            want.append(-oracle.gen_teacher_forced(og, inp, tgt)["logp"].astype(np.float64).mean())
        for precision in GEN_PRECISIONS:
            got = tg.nll.cal_nll(gen, batches, precision=precision)
            assert isinstance(got, float) and got == pytest.approx(round(float(np.mean(want)), 4), abs=1.01e-4)


@pytest.mark.parametrize("B,R,p", [(1, 1, 0.25), (2, 5, 0.5), (33, 3, 0.0)])
def test_mali_reward_edges(B, R, p):
    """get_mali_reward with one sample, one mask, no dropout at all -- against the oracle (B = 1 normalises a single
    value by itself: the reference's formula gives 0 there, and so must this)."""
    V, L = 60, 20
    sd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=4)
    dis = tg.modules.load_numpy_state(tg.modules.SeqGAN_D(64, V, dropout=p), sd, DEV).train()
    samples = synth.tokens(91, (B, L), V)
    du = synth.uniform01(92, (R, B, dis.feature_dim))
    rew = tg.maligan.get_mali_reward(dis, _t(samples), R, dropout_uniforms=_t(du)).cpu().numpy()
    want = oracle.mali_reward(oracle.Dis(sd), samples, R, p, du)
    assert rew.shape == (B, L)
    np.testing.assert_allclose(rew, want, atol=2e-4 / B, rtol=2e-4)


def test_get_token_reward_against_oracle():
    """ROLLOUT.get_token_reward (utils/rollout.py:186-204): the mean over rollout_num completions from ONE given_num."""
    c = _cases.load("reward_tiny_flat")
    gen, dis = _gen(c), _dis(c).eval()
    sent, u, _ = _cases.reward_inputs(c)
    R, L, B = int(c["R"]), int(c["L"]), int(c["B"])
    og, od = oracle.Gen(_cases.gen_sd(c)), oracle.Dis(_cases.dis_sd(c))
    full = oracle.gen_rollout(og, sent, R, u)                    # [R, L, B, L]
    for g in (1, 7, L):
        want = np.stack([oracle.dis_forward(od, full[i, g - 1])["reward"] for i in range(R)]).mean(0)
        got = tg.ROLLOUT(gen).get_token_reward(_t(sent), R, dis, 0, g, uniforms=_t(u))
        assert tuple(got.shape) == (B,)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-4, rtol=0)


def test_empty_batches_are_no_ops():
    """Empty inputs behave as they do in the reference (ATen handles a zero-row batch): empty outputs of the right
    shape, no error -- scorer, get_reward with no sentence, NLL, sample(0), MaliGAN reward, policy-gradient loss."""
    V, L = 50, 20
    gen = tg.modules.synthetic_generator(V, 32, 32, L, device=DEV)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=DEV)
    e = torch.zeros(0, L, dtype=torch.long, device=DEV)
    r = tg.ops.dis_score(dis, e, want_logits=True, want_feature=True)
    assert tuple(r["reward"].shape) == (0,) and tuple(r["logits"].shape) == (0, 2) and tuple(r["feature"].shape) == (0, 1720)
    assert tuple(tg.discriminator.forward(dis, e).shape) == (0, 2)
    assert tuple(tg.ROLLOUT(gen).get_reward(e, 2, dis).shape) == (0, L)
    assert tuple(tg.ops.gen_nll(gen, e, e).shape) == (0, L)
    assert tuple(tg.generator.sample(gen, 0, 4, 1).shape) == (0, L)
    assert tuple(tg.maligan.get_mali_reward(dis, e, 2).shape) == (0, L)
    loss = tg.pgloss.batchPGLoss(gen, e, e, torch.zeros(0, L, device=DEV))
    assert loss.item() == 0.0
    with pytest.raises(RuntimeError):          # rollout_num 0: the reference divides by it (NaN rewards); refused here
        tg.ROLLOUT(gen).get_reward(torch.zeros(3, L, dtype=torch.long, device=DEV), 0, dis)


def test_out_of_range_tokens_are_clamped_not_faulted():
    """Token ids outside [0, V) -- the reference raises IndexError from the embedding -- are clamped to the nearest valid
    id for memory safety: the call completes and equals the call on the clamped sentences."""
    V, L, B, R = 40, 20, 6, 2
    gen = tg.modules.synthetic_generator(V, 32, 32, L, device=DEV)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=DEV).eval()
    sent = synth.tokens(3, (B, L), V)
    bad = sent.copy()
    bad[0, 0], bad[1, 5], bad[2, 19], bad[3, 7] = -1, V, V + 1000, -(2 ** 40)
    u = _t(synth.uniform01(4, (R, L, L, B)))
    want = tg.ROLLOUT(gen).get_reward(_t(np.clip(bad, 0, V - 1)), R, dis, uniforms=u)
    got = tg.ROLLOUT(gen).get_reward(_t(bad), R, dis, uniforms=u)
    assert torch.isfinite(got).all() and torch.equal(got, want)
    a = tg.ops.dis_score(dis, _t(bad))["reward"]
    b = tg.ops.dis_score(dis, _t(np.clip(bad, 0, V - 1)))["reward"]
    assert torch.equal(a, b)


def test_strided_and_int32_token_inputs():
    """Sentences given as a strided view or as int32 (a loader's dtype) give the result of the contiguous int64 call."""
    V, L, B, R = 40, 20, 6, 2
    gen = tg.modules.synthetic_generator(V, 32, 32, L, device=DEV)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=0, device=DEV).eval()
    sent = _t(synth.tokens(3, (B, L), V))
    u = _t(synth.uniform01(4, (R, L, L, B)))
    want = tg.ROLLOUT(gen).get_reward(sent, R, dis, uniforms=u)
    wide = torch.zeros(B, 2 * L, dtype=torch.long, device=DEV)
    wide[:, ::2] = sent
    view = wide[:, ::2]
    assert not view.is_contiguous()
    assert torch.equal(tg.ROLLOUT(gen).get_reward(view, R, dis, uniforms=u), want)
    assert torch.equal(tg.ROLLOUT(gen).get_reward(sent.int(), R, dis, uniforms=u), want)
    assert torch.equal(tg.discriminator.forward(dis, view), tg.discriminator.forward(dis, sent))
    assert torch.equal(tg.ops.gen_nll(gen, view, sent.int()), tg.ops.gen_nll(gen, sent, sent))


def test_adversarial_generator_steps_like_the_instructor():
    """SeqGANInstructor.adv_train_generator (instructor/oracle_data/seqgan_instructor.py:97-114) on the drop-in
    surfaces alone: sample -> prepare (start letter prepended) -> ROLLOUT.get_reward -> batchPGLoss -> optimizer step,
    three times.  The generator is stepped between calls: the same uniforms must give other completions afterwards
    (parameters are read on every call), rewards stay probabilities, nothing goes non-finite."""
    V, L, B, R = 300, 20, 16, 4
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L, name="seqgan"),
                                      synth.gen_state(V, 32, 32, seed=2, std=0.5), DEV)
    dis = tg.modules.synthetic_discriminator("seqgan", V, seed=1, device=DEV)
    opt = torch.optim.Adam(gen.parameters(), lr=1e-2)
    u = _t(synth.uniform01(5, (R, L, L, B)))
    us = _t(synth.uniform01(6, (L, B)))
    seen = []
    for step in range(3):
        target = tg.generator.sample(gen, B, B, 1, uniforms=us).to(DEV)           # CPU int64 [B, L] -> device
        inp = torch.cat([torch.ones(B, 1, dtype=torch.long, device=DEV), target[:, :-1]], 1)  # GenDataIter.prepare
        ro = tg.ROLLOUT(gen, True)                                               # a new ROLLOUT per step, as the instructor
        rewards, raw, comp = ro.get_reward(target, R, dis, uniforms=u, return_parts=True)
        assert tuple(rewards.shape) == (B, L) and bool(((raw >= 0) & (raw <= 1)).all())
        loss = tg.pgloss.batchPGLoss(gen, inp, target, rewards)
        assert torch.isfinite(loss)
        opt.zero_grad()
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in gen.parameters())
        opt.step()
        seen.append((target.clone(), comp[:, 0].clone()))
    assert not torch.equal(seen[0][0], seen[2][0])      # the stepped generator samples other sentences
    assert not torch.equal(seen[0][1], seen[2][1])      # ... and rolls out other completions from the same uniforms


def test_concurrent_calls_on_two_streams():
    """SURVEY 8(b) threading contract: the entry points are re-entrant per stream.  Two threads, each on a stream of
    its own, run get_reward with different uniforms at the same time, several times over; each result must equal the
    same call made alone."""
    import threading
    c = _cases.load("reward_tiny_dropout")
    gen, dis = _gen(c), _dis(c)
    sent, u, du = _cases.reward_inputs(c)
    R, col, p = int(c["R"]), int(c["current_k"]), float(c["dropout_p"])
    us = [_t(u), _t(synth.uniform01(991, u.shape))]
    dus = [_t(du), _t(synth.uniform01(992, du.shape))]
    sent_t = _t(sent)
    alone = [tg.ROLLOUT(gen).get_reward(sent_t, R, dis, col, uniforms=us[k], dropout_uniforms=dus[k]).clone() for k in (0, 1)]
    torch.cuda.synchronize()
    outs, errs = [[], []], []

    def work(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ro = tg.ROLLOUT(gen)
                for _ in range(8):
                    outs[k].append(ro.get_reward(sent_t, R, dis, col, uniforms=us[k], dropout_uniforms=dus[k]))
            st.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in (0, 1):
        assert len(outs[k]) == 8
        for r in outs[k]:
            assert torch.equal(r, alone[k])
    assert not torch.equal(alone[0], alone[1])


def test_errors_and_cpu_inputs():
    gen = tg.modules.synthetic_generator(100, 32, 32, 20, device=DEV)
    with pytest.raises(RuntimeError):
        tg.ROLLOUT(gen).get_reward(torch.zeros(4, 20, dtype=torch.long), 2, None)  # CPU tensor
    bad = tg.modules.LSTMGenerator(48, 96, 100, 20).to(DEV)  # H not in {32,64,128,256,512}
    with pytest.raises(RuntimeError):
        tg.ops.gen_sample(bad, 4, 20, 1, torch.rand(20, 4, device=DEV))


# ------------------------------------------------------------------ policy-gradient loss
def _pg_gen(c, sd):
    g = tg.modules.LSTMGenerator(int(c.get("E", 32)), int(c.get("H", 32)), int(c["V"]), int(c["L"]),
                                 padding_idx=int(c["padding_idx"]), name=str(c["model"]))
    return tg.modules.load_numpy_state(g, sd, DEV)


def _grads_of(gen):
    return {k: p.grad.detach().cpu().numpy() for k, p in gen.named_parameters()}


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
@pytest.mark.parametrize("name", _cases.names("pg_loss"))
def test_pg_loss_and_gradients(name, precision):
    """pgloss.batchPGLoss / adv_loss + .backward() against the reference's loss and .grad (golden) and the oracle.
    precision: beyond E = H = 32 the three vocabulary products run as split-fp16 x3 GEMMs or exact fp32 ones (the fused
    E = H = 32 kernels are fp32 either way); the same 5e-5 holds in both."""
    c = _cases.load(name)
    sd, inp, target, reward, senti, weight = _cases.pg_inputs(c)
    gen = _pg_gen(c, sd)
    fn = tg.pgloss.adv_loss if str(c["model"]) == "maligan" else tg.pgloss.batchPGLoss
    loss = fn(gen, _t(inp), _t(target), _t(reward), precision=precision)
    gen.zero_grad()
    loss.backward()
    assert loss.item() == pytest.approx(float(c["loss"]), rel=1e-5)
    got = _grads_of(gen)
    _cases.assert_grads_close(got, {k: c["grad." + k] for k in _cases.PG_KEYS}, rel=5e-5, what=name + " vs reference")
    oloss, ograds = oracle.pg_loss_grads(oracle.Gen(sd), inp, target, weight, use_prob=senti,
                                         padding_idx=int(c["padding_idx"]))
    assert loss.item() == pytest.approx(oloss, rel=1e-5)
    _cases.assert_grads_close(got, ograds, rel=5e-5, what=name + " vs oracle")
    assert not got["embeddings.weight"][int(c["padding_idx"])].any()


def test_pg_loss_c2_shape_against_oracle():
    """Adversarial-step shape of the real-data config (V5255 L51 B64): gradients against the oracle, an optimizer
    step through the returned autograd node, ragged row / vocabulary tails (64*51 and 5255 are not multiples of 32)."""
    V, L, B = 5255, 51, 64
    sd = synth.gen_state(V, 32, 32, seed=0, std=0.1)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(32, 32, V, L, name="seqgan"), sd, DEV)
    target = synth.tokens(31, (B, L), V)
    inp = np.concatenate([np.ones((B, 1), np.int64), target[:, :-1]], 1)
    reward = synth.uniform01(32, (B, L)).astype(np.float32)
    opt = torch.optim.Adam(gen.parameters(), lr=1e-2)
    loss = tg.pgloss.batchPGLoss(gen, _t(inp), _t(target), _t(reward))
    opt.zero_grad()
    loss.backward()
    oloss, ograds = oracle.pg_loss_grads(oracle.Gen(sd), inp, target, reward, padding_idx=0)
    assert loss.item() == pytest.approx(oloss, rel=1e-5)
    _cases.assert_grads_close(_grads_of(gen), ograds, rel=5e-5, what="c2")
    before = gen.lstm2out.weight.detach().clone()
    opt.step()
    assert (gen.lstm2out.weight.detach() - before).abs().max().item() > 1e-3
    loss2 = tg.pgloss.batchPGLoss(gen, _t(inp), _t(target), _t(reward))
    assert loss2.item() < loss.item()  # one Adam step on the same batch lowers the loss


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
@pytest.mark.parametrize("V,E,H,B,L", [(2, 32, 32, 1, 2), (5, 32, 32, 33, 7), (129, 32, 32, 2, 20), (7, 64, 96, 1, 3),
                                       (33, 96, 32, 35, 4), (4099, 64, 64, 9, 5), (515, 32, 128, 70, 6)])
def test_pg_loss_edge_shapes_against_oracle(V, E, H, B, L, precision):
    """The policy-gradient loss and all seven gradients on the smallest and most ragged batches (one sentence of two
    tokens, a two-word vocabulary, row counts off every tile size), fused kernels (E = H = 32) and GEMM path alike."""
    sd = synth.gen_state(V, E, H, seed=V + L, std=0.3)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L, name="seqgan"), sd, DEV)
    target = synth.tokens(41, (B, L), V)
    inp = np.concatenate([np.zeros((B, 1), np.int64), target[:, :-1]], 1)   # token 0 = padding_idx: its embedding gets no gradient
    reward = synth.uniform01(42, (B, L)).astype(np.float32)
    loss = tg.pgloss.batchPGLoss(gen, _t(inp), _t(target), _t(reward), precision=precision)
    gen.zero_grad()
    loss.backward()
    oloss, ograds = oracle.pg_loss_grads(oracle.Gen(sd), inp, target, reward, padding_idx=0)
    assert loss.item() == pytest.approx(oloss, rel=1e-5)
    _cases.assert_grads_close(_grads_of(gen), ograds, rel=5e-5, what="edge %s" % ((V, E, H, B, L),))
    assert not gen.embeddings.weight.grad[0].any()


def test_pg_loss_rejects_unsupported_shapes_in_forward():
    gen = tg.modules.LSTMGenerator(48, 80, 100, 20).to(DEV)   # E, H not multiples of 32
    t = _t(synth.tokens(1, (4, 20), 100))
    with pytest.raises(RuntimeError, match="not supported"):   # in forward, before any work is done
        tg.pgloss.batchPGLoss(gen, t, t, torch.ones(4, 20, device=DEV))


@pytest.mark.parametrize("precision", GEN_PRECISIONS)
def test_pg_loss_wide_shape_against_oracle_and_kernels_used(precision):
    """A shape with ragged tiles everywhere (V, N*L, E, 4H not multiples of 128; N*L*V > one GEMM tile in every
    dimension) through the GEMM path: loss and gradients against the oracle, the fused E = H = 32 kernels untouched."""
    V, E, H, L, B = 1531, 96, 160, 9, 37
    sd = synth.gen_state(V, E, H, seed=41, std=0.2)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L, name="seqgan"), sd, DEV)
    target = synth.tokens(42, (B, L), V)
    inp = np.concatenate([np.ones((B, 1), np.int64), target[:, :-1]], 1)
    reward = synth.uniform01(43, (B, L)).astype(np.float32)
    tg._lib.profile_enable(True)
    tg._lib.profile_read()
    loss = tg.pgloss.batchPGLoss(gen, _t(inp), _t(target), _t(reward), precision=precision)
    gen.zero_grad()
    loss.backward()
    prof = tg._lib.profile_read()
    tg._lib.profile_enable(False)
    # the vocabulary products carry their own profile id: logits (forward), logits again + dh_proj + dW_out (backward)
    assert prof["pgw_forward"][1] >= 1 and prof["pgw_backward"][1] >= 1 and prof["pg_backward"][1] == 0
    assert prof["pgw_proj"][1] == 2 and prof["pgw_proj_dh"][1] == 1 and prof["pgw_proj_dw"][1] == 1
    oloss, ograds = oracle.pg_loss_grads(oracle.Gen(sd), inp, target, reward, padding_idx=0)
    assert loss.item() == pytest.approx(oloss, rel=1e-5)
    _cases.assert_grads_close(_grads_of(gen), ograds, rel=5e-5, what="wide")


def test_pg_loss_split_across_logit_chunks_against_oracle():
    """The GEMM path keeps at most 1 GiB of logits alive: V 32 000 with 10 200 token rows is two chunks (8 192 + 2 008
    rows), so dW_out and db_out accumulate over chunks and dh_proj is cut over k (32 000) into slices summed in order.
    Loss and all seven gradients against the oracle in the default split-fp16 mode; a second call is bit-identical
    but for the embedding gradient (float atomics)."""
    V, E, H, L, B = 32000, 32, 64, 51, 200
    sd = synth.gen_state(V, E, H, seed=43, std=0.1)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L, name="seqgan"), sd, DEV)
    target = synth.tokens(44, (B, L), V)
    inp = np.concatenate([np.ones((B, 1), np.int64), target[:, :-1]], 1)
    reward = synth.uniform01(45, (B, L)).astype(np.float32)
    runs = []
    for _ in range(2):
        loss = tg.pgloss.batchPGLoss(gen, _t(inp), _t(target), _t(reward))
        gen.zero_grad()
        loss.backward()
        runs.append((loss.item(), {k: p.grad.detach().clone() for k, p in gen.named_parameters()}))
    assert runs[0][0] == runs[1][0]
    for k in _cases.PG_KEYS:
        if k != "embeddings.weight":
            assert torch.equal(runs[0][1][k], runs[1][1][k]), k
    oloss, ograds = oracle.pg_loss_grads(oracle.Gen(sd), inp, target, reward, padding_idx=0)
    assert runs[0][0] == pytest.approx(oloss, rel=1e-5)
    _cases.assert_grads_close({k: v.cpu().numpy() for k, v in runs[0][1].items()}, ograds, rel=5e-5, what="two chunks")


@pytest.mark.parametrize("V,E,H,N,L,std", [(900, 96, 160, 33, 5, 0.2), (600, 32, 768, 5, 5, 0.05), (333, 64, 96, 130, 7, 0.3)])
def test_teacher_forced_beyond_the_persistent_kernels(V, E, H, N, L, std):
    """LSTMGenerator.forward (logits, log-probabilities, states) for hidden sizes outside {32, 64, 128, 256, 512}:
    tg_gen_teacher_forced_ws runs them as exact-fp32 GEMMs; 1e-5 against the oracle like every other shape."""
    gsd = synth.gen_state(V, E, H, seed=16, std=std)
    gen = tg.modules.load_numpy_state(tg.modules.LSTMGenerator(E, H, V, L), gsd, DEV)
    og = oracle.Gen(gsd)
    assert tg._lib.lib().tg_gen_supported(V, E, H) == 1
    tgt = synth.tokens(27, (N, L), V)
    inp = np.concatenate([np.ones((N, 1), np.int64), tgt[:, :-1]], 1)
    o = oracle.gen_teacher_forced(og, inp, tgt, want_logits=True, want_states=True)
    r = tg.ops.gen_teacher_forced(gen, _t(inp), _t(tgt), want_lse=True, want_logits=True, want_states=True)
    np.testing.assert_allclose(r["logits"].cpu().numpy(), o["logits"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(r["lse"].cpu().numpy(), o["lse"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["logp"].cpu().numpy(), o["logp"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["states"].cpu().numpy(), o["states"], rtol=1e-4, atol=5e-5)
    # without logits_out / states_out the chunked path and the library's own state scratch are used
    r2 = tg.ops.gen_teacher_forced(gen, _t(inp), _t(tgt), want_lse=True)
    assert torch.equal(r2["logp"], r["logp"]) and torch.equal(r2["lse"], r["lse"])
    lp = tg.generator.forward_logprobs(gen, _t(inp)).view(N, L, -1).cpu().numpy()
    np.testing.assert_allclose(lp, o["logits"] - o["lse"][..., None], rtol=1e-5, atol=2e-5)
