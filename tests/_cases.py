"""Rebuild golden-case inputs from the seeds stored in tests/golden/*.npz."""
import glob
import os

import numpy as np

import textgan_pytorch_amd as tg

synth = tg.synth
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


def names(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        z = np.load(p, allow_pickle=False)
        if str(z["kind"]) == kind:
            out.append(os.path.basename(p)[:-4])
    return out


def gen_sd(c):
    return synth.gen_state(int(c["V"]), int(c["E"]), int(c["H"]), seed=int(c["gseed"]), std=float(c["gstd"]))


def dis_sd(c):
    if str(c["dis_kind"]) == "seqgan":
        return synth.dis_state(int(c["V"]), 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=int(c["dseed"]))
    return synth.dis_state(int(c["V"]), 64, *synth.SENTIGAN_D_FILTERS, n_out=int(c["k_label"]) + 1,
                           seed=int(c["dseed"]))


def reward_inputs(c):
    V, L, B, R = int(c["V"]), int(c["L"]), int(c["B"]), int(c["R"])
    sent = synth.tokens(int(c["sseed"]), (B, L), V)
    u = synth.uniform01(int(c["useed"]), (R, L, L, B))
    du = None
    if int(c["dropout"]):
        F = sum(synth.SEQGAN_D_FILTERS[1] if str(c["dis_kind"]) == "seqgan" else synth.SENTIGAN_D_FILTERS[1])
        du = synth.uniform01(int(c["useed"]) + 7, (R * L, B, F))
    return sent, u, du


def sampled_mask(R, L, B):
    """True where completions[i, g-1, b, t] is a sampled position (t >= g)."""
    g = np.arange(1, L + 1)[:, None]
    t = np.arange(L)[None, :]
    return np.broadcast_to((t >= g)[None, :, None, :], (R, L, B, L))


#: largest |cdf - u| seen at a token disagreement, per test id (written to gpurun_out/tie_stats.json at session end)
TIE_STATS = {}


def check_token_parity(got_replay, ref, uniforms, gen_oracle, min_agree=0.999, tie_tol=1e-6, max_check=1000, what=None):
    """Step-wise (replay-mode) token parity, north_star's criterion: >= 99.9 % of
    sampled positions identical, and every disagreement a CDF tie within fp32
    rounding.  got_replay / ref: [R,L,B,L] completions, got_replay produced with
    forced=ref; uniforms [R,L,L,B].

    tie_tol: the disagreeing tokens must be neighbours in CDF order with every boundary between them within
    tie_tol of u.  1e-6 is what fp32 leaves at a boundary: the cumulative probability there is a sum of V terms
    exp(logit - max) whose logits carry ~1e-6 relative error (H-term fp32 dot products of magnitude up to ~15 with
    N(0,1) weights), far above the 1.2e-7 of a single fp32 rounding, far below the 2e-5 accepted in round 1.  Every
    disagreement is inspected (up to max_check = 1000 per call; more than that fails min_agree on any case here)."""
    from oracle import oracle
    R, L, B, _ = ref.shape
    m = sampled_mask(R, L, B)
    diff = (got_replay != ref) & m
    agree = 1.0 - diff.sum() / max(1, m.sum())
    assert agree >= min_agree, "step-wise token agreement %.6f" % agree
    assert np.array_equal(got_replay[~m], ref[~m]), "prefix tokens must be copied verbatim"
    idx = np.argwhere(diff)

    def logits_of(i, gi, b, t):
        row = ref[i, gi, b].astype(np.int64)
        r = oracle.gen_teacher_forced(gen_oracle, row[None, :], want_logits=True)
        return r["logits"][0, t - 1]

    _check_ties(idx, got_replay, ref, uniforms, logits_of, tie_tol, max_check, int(m.sum()), what)
    return agree


def _check_ties(idx, got, ref, uniforms, logits_of, tie_tol, max_check, draws, what):
    """Every disagreement got != ref at idx = (i, j, b, t) must be a CDF tie: the two tokens are neighbours in CDF
    order and every boundary between them lies within tie_tol of the draw's uniform u[i, j, t, b].  logits_of(i, j,
    b, t) returns the oracle's logits the draw was taken from.  Records the session's statistics in TIE_STATS."""
    import os
    assert len(idx) <= max_check, "%d disagreements: more than the %d this check inspects" % (len(idx), max_check)
    worst = 0.0
    for (i, gi, b, t) in idx:
        lg = logits_of(i, gi, b, t).astype(np.float64)
        w = np.exp(lg - lg.max())
        cdf = np.cumsum(w) / w.sum()
        a, c = int(got[i, gi, b, t]), int(ref[i, gi, b, t])
        u = float(uniforms[i, gi, t, b])
        lo, hi = min(a, c), max(a, c)
        # every boundary between the two tokens must sit within tie_tol of u
        gap = float(np.abs(cdf[lo:hi] - u).max())
        assert gap <= tie_tol, (i, gi + 1, b, t, a, c, u, cdf[lo:hi])
        worst = max(worst, gap)
    key = what or os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    st = TIE_STATS.setdefault(key, {"draws": 0, "disagreements": 0, "max_tie_gap": 0.0})
    st["draws"] += draws
    st["disagreements"] += int(len(idx))
    st["max_tie_gap"] = max(st["max_tie_gap"], worst)


def check_sample_token_parity(got_replay, ref, uniforms, gen_oracle, start_letter=1, min_agree=0.999, tie_tol=1e-6,
                              max_check=1000, what=None):
    """check_token_parity for LSTMGenerator.sample (models/generator.py:62-84): got_replay / ref [N,L] (got_replay
    produced with forced=ref), uniforms [L,N]; every position is a draw, position t from the logits after feeding
    [start_letter, ref[:t]].  >= 99.9 % identical and every disagreement a CDF tie within tie_tol."""
    from oracle import oracle
    N, L = ref.shape
    diff = got_replay != ref
    agree = 1.0 - diff.sum() / max(1, diff.size)
    assert agree >= min_agree, "step-wise token agreement %.6f" % agree

    def logits_of(i, gi, n, t):
        inp = np.concatenate([[start_letter], ref[n, :-1]]).astype(np.int64)
        return oracle.gen_teacher_forced(gen_oracle, inp[None, :], want_logits=True)["logits"][0, t]

    idx = np.array([(0, 0, n, t) for n, t in np.argwhere(diff)], dtype=np.int64).reshape(-1, 4)
    _check_ties(idx, got_replay[None, None], ref[None, None], uniforms[None, None], logits_of, tie_tol, max_check,
                int(diff.size), what)
    return agree


# ------------------------------------------------------------------ LeakGAN rollout
def leak_inputs(c):
    """(LeakGAN_G state_dict, LeakGAN_D state_dict, sentences [B,L], uniforms [R,T,L,B], dropout uniforms or None)
    of a leak_reward golden (tools/make_goldens_leak.py)."""
    V, E, H, K, G, L, B, R, step = [int(c[k]) for k in ("V", "E", "H", "K", "G", "L", "B", "R", "step")]
    gsd = synth.leak_state(V, E, H, K, G, B, seed=int(c["gseed"]), std=float(c["gstd"]))
    dsd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=int(c["dseed"]))
    T = L // step
    sent = synth.tokens(int(c["sseed"]), (B, L), V)
    u = synth.uniform01(int(c["useed"]), (R, T, L, B))
    du = synth.uniform01(int(c["useed"]) + 7, (R * T, B, G)) if int(c["dropout"]) else None
    return gsd, dsd, sent, u, du


def leak_sampled_mask(R, T, B, L, step):
    """True where completions[i, t, b, pos] is a sampled position (pos >= given_num = t*step + 1)."""
    g = (np.arange(T) * step + 1)[:, None]
    pos = np.arange(L)[None, :]
    return np.broadcast_to((pos >= g)[None, :, None, :], (R, T, B, L))


def check_leak_token_parity(got_replay, ref, uniforms, leak_oracle, dis_oracle, min_agree=0.999, tie_tol=1e-6,
                            max_check=1000, what=None):
    """check_token_parity for rollout_mc_search_leakgan's completions [R,T,B,L] (uniforms [R,T,L,B])."""
    from oracle import oracle
    R, T, B, L = ref.shape
    m = leak_sampled_mask(R, T, B, L, leak_oracle.step_size)
    diff = (got_replay != ref) & m
    agree = 1.0 - diff.sum() / max(1, m.sum())
    assert agree >= min_agree, "step-wise token agreement %.6f" % agree
    assert np.array_equal(got_replay[~m], ref[~m]), "prefix tokens must be copied verbatim"

    def logits_of(i, t, b, pos):
        return oracle.leak_logits_at(leak_oracle, dis_oracle, ref[i, t, b].astype(np.int64), b, pos - 1)

    _check_ties(np.argwhere(diff), got_replay, ref, uniforms, logits_of, tie_tol, max_check, int(m.sum()), what)
    return agree


PG_KEYS = ("embeddings.weight", "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
           "lstm2out.weight", "lstm2out.bias")


def pg_inputs(c):
    """(gen state_dict, inp, target, reward, use_prob, weight) of a pg_loss golden (tools/make_goldens_pg.py)."""
    V, L, B = int(c["V"]), int(c["L"]), int(c["B"])
    E, H = int(c.get("E", 32)), int(c.get("H", 32))   # goldens of round 1 carry no E / H: the reference's 32
    sd = synth.gen_state(V, E, H, seed=int(c["gseed"]), std=float(c["gstd"]))
    target = synth.tokens(int(c["sseed"]), (B, L), V)
    inp = np.concatenate([np.full((B, 1), int(c["start_letter"]), np.int64), target[:, :-1]], 1)
    reward = synth.uniform01(int(c["rseed"]), (B, L)).astype(np.float32)
    senti = str(c["model"]) == "sentigan"
    return sd, inp, target, reward, senti, (1.0 - reward if senti else reward)


def assert_grads_close(got, want, rel=2e-5, what=""):
    """Per tensor: max |got - want| <= rel * max |want| (gradients of one tensor share a scale)."""
    for k in PG_KEYS:
        g, w = np.asarray(got[k], np.float64), np.asarray(want[k], np.float64)
        if g.shape != w.shape and g.ndim == 2 and w.shape[1] == g.shape[1] and w.shape[0] < g.shape[0]:
            g = g[::-(-g.shape[0] // w.shape[0])]   # the fixture keeps every n-th row of a large matrix (make_goldens_pg)
        assert g.shape == w.shape, (k, g.shape, w.shape)
        err, scale = np.abs(g - w).max(), np.abs(w).max()
        assert err <= rel * scale + 1e-9, "%s %s: max err %.3e vs scale %.3e" % (what, k, err, scale)
