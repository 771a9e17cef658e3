"""ctypes front end of the CPU oracle (oracle/textgan_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of textgan_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Inputs are numpy state_dicts (the reference's parameter names) and numpy
arrays; outputs are numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtextgan_oracle.so")
MAX_FILTERS = 16


def build(force=False):
    src = os.path.join(_HERE, "textgan_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtextgan_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class GenW(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out")] + \
               [("V", C.c_int32), ("E", C.c_int32), ("H", C.c_int32), ("_pad", C.c_int32)]


class DisW(C.Structure):
    _fields_ = [("emb", C.c_void_p), ("conv_w", C.c_void_p * MAX_FILTERS), ("conv_b", C.c_void_p * MAX_FILTERS),
                ("hw_w", C.c_void_p), ("hw_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
                ("f", C.c_int32 * MAX_FILTERS), ("n", C.c_int32 * MAX_FILTERS), ("n_filters", C.c_int32),
                ("V", C.c_int32), ("Ed", C.c_int32), ("F", C.c_int32), ("n_out", C.c_int32), ("precision", C.c_int32)]


class LeakW(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("emb", "w_wih", "w_whh", "w_bih", "w_bhh", "m_wih", "m_whh", "m_bih", "m_bhh",
                                          "w2g_w", "w2g_b", "m2g_w", "m2g_b", "g2g_w", "goal_init")] + \
               [(k, C.c_int32) for k in ("V", "E", "H", "G", "K", "step_size", "B_init", "_pad")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_span_len.restype = C.c_int
        _lib.orc_group_tiles.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_dis_forward.restype = C.c_int
        _lib.orc_get_reward.restype = C.c_int
        _lib.orc_leak_rollout.restype = C.c_int
        _lib.orc_get_reward_leakgan.restype = C.c_int
        _lib.orc_leak_logits_at.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Gen:
    """Holds fp32 copies of an LSTMGenerator state_dict (models/generator.py:33-35)."""

    def __init__(self, sd):
        g = lambda k: _f32(np.asarray(sd[k]))
        self.a = [g("embeddings.weight"), g("lstm.weight_ih_l0"), g("lstm.weight_hh_l0"), g("lstm.bias_ih_l0"),
                  g("lstm.bias_hh_l0"), g("lstm2out.weight"), g("lstm2out.bias")]
        self.V, self.E = self.a[0].shape
        self.H = self.a[2].shape[1]
        self.w = GenW(*[x.ctypes.data for x in self.a], self.V, self.E, self.H, 0)


class Dis:
    """Holds fp32 copies of a CNNDiscriminator state_dict (models/discriminator.py:30-36)."""

    def __init__(self, sd):
        g = lambda k: _f32(np.asarray(sd[k]))
        self.keep = []
        w = DisW()
        emb = g("embeddings.weight"); self.keep.append(emb)
        w.emb = emb.ctypes.data
        self.V, self.Ed = emb.shape
        k = 0
        F = 0
        while "convs.%d.weight" % k in sd:
            cw = g("convs.%d.weight" % k); cb = g("convs.%d.bias" % k)
            n, _, f, ed = cw.shape
            cw = np.ascontiguousarray(cw.reshape(n, f * ed))
            self.keep += [cw, cb]
            w.conv_w[k] = cw.ctypes.data; w.conv_b[k] = cb.ctypes.data
            w.f[k] = f; w.n[k] = n
            F += n; k += 1
        w.n_filters = k
        for name, key in (("hw_w", "highway.weight"), ("hw_b", "highway.bias"),
                          ("out_w", "feature2out.weight"), ("out_b", "feature2out.bias")):
            arr = g(key); self.keep.append(arr); setattr(w, name, arr.ctypes.data)
        self.F = F
        self.n_out = self.keep[-1].shape[0]
        w.V, w.Ed, w.F, w.n_out = self.V, self.Ed, F, self.n_out
        self.w = w


class Leak:
    """Holds fp32 copies of a LeakGAN_G state_dict (models/LeakGAN_G.py:40-48) and its step_size."""

    KEYS = ("embeddings.weight", "worker.weight_ih_l0", "worker.weight_hh_l0", "worker.bias_ih_l0", "worker.bias_hh_l0",
            "manager.weight_ih_l0", "manager.weight_hh_l0", "manager.bias_ih_l0", "manager.bias_hh_l0",
            "work2goal.weight", "work2goal.bias", "mana2goal.weight", "mana2goal.bias", "goal2goal.weight", "goal_init")

    def __init__(self, sd, step_size=4):
        self.a = [_f32(np.asarray(sd[k])) for k in self.KEYS]
        self.V, self.E = self.a[0].shape
        self.H = self.a[2].shape[1]
        self.K, self.G = self.a[13].shape
        self.B_init = self.a[14].shape[0]
        self.step_size = int(step_size)
        assert self.a[9].shape == (self.V * self.K, self.H) and self.a[5].shape == (4 * self.H, self.G)
        self.w = LeakW(*[x.ctypes.data for x in self.a], self.V, self.E, self.H, self.G, self.K, self.step_size,
                       self.B_init, 0)


def num_threads():
    return lib().orc_num_threads()


def span_len(V):
    return lib().orc_span_len(int(V))


def group_tiles(V):
    return lib().orc_group_tiles(int(V))


def sample_from_probs(p, u):
    """Canonical inverse-CDF draw; the golden harness's torch.multinomial(p,1)."""
    p = _f32(p); u = _f32(u)
    N, V = p.shape
    tok = np.empty(N, np.int64)
    lib().orc_sample_from_probs(_p(p), _p(u), N, V, _p(tok))
    return tok


def gen_teacher_forced(gen, inp, target=None, want_logits=False, want_states=False):
    inp = _i64(inp)
    N, L = inp.shape
    tgt = None if target is None else _i64(target)
    logp = np.empty((N, L), np.float32) if tgt is not None else None
    lse = np.empty((N, L), np.float32)
    logits = np.empty((N, L, gen.V), np.float32) if want_logits else None
    states = np.empty((L + 1, N, 2, gen.H), np.float32) if want_states else None
    lib().orc_gen_teacher_forced(C.byref(gen.w), _p(inp), _p(tgt), N, L, _p(logp), _p(lse), _p(logits), _p(states))
    return {"logp": logp, "lse": lse, "logits": logits, "states": states}


def gen_rollout(gen, sentences, R, uniforms, i_lo=0, i_hi=None, forced=None):
    s = _i64(sentences)
    B, L = s.shape
    i_hi = R if i_hi is None else i_hi
    u = _f32(uniforms)
    assert u.size == R * L * L * B
    comp = np.zeros((R, L, B, L), np.int64)
    f = None if forced is None else _i64(forced)
    lib().orc_gen_rollout(C.byref(gen.w), _p(s), B, L, R, i_lo, i_hi, _p(u), _p(f), _p(comp))
    return comp


def gen_sample(gen, N, L, start_letter, uniforms, forced=None):
    u = _f32(uniforms)
    assert u.size == L * N
    out = np.empty((N, L), np.int64)
    f = None if forced is None else _i64(forced)
    lib().orc_gen_sample(C.byref(gen.w), N, L, C.c_int64(start_letter), _p(u), _p(f), _p(out))
    return out


def dis_forward(dis, seqs, col=1, dropout_p=0.0, dropout_uniforms=None, want_feature=False):
    s = _i64(seqs)
    N, L = s.shape
    du = None if dropout_uniforms is None else _f32(dropout_uniforms)
    reward = np.empty(N, np.float32)
    logits = np.empty((N, dis.n_out), np.float32)
    feat = np.empty((N, dis.F), np.float32) if want_feature else None
    rc = lib().orc_dis_forward(C.byref(dis.w), _p(s), N, L, col, C.c_float(dropout_p), _p(du),
                               _p(reward), _p(logits), _p(feat))
    if rc:
        raise RuntimeError("oracle: sequence length %d shorter than a filter width" % L)
    return {"reward": reward, "logits": logits, "feature": feat}


def reward_reduce(raw, B, L, R):
    raw = _f32(raw)
    out = np.empty((B, L), np.float32)
    lib().orc_reward_reduce(_p(raw), B, L, R, _p(out))
    return out


def get_reward(gen, dis, sentences, R, uniforms, col=1, dropout_p=0.0, dropout_uniforms=None,
               i_lo=0, i_hi=None, reduce=True):
    """ROLLOUT.get_reward (utils/rollout.py:128-154). Returns dict(completions, raw, reward)."""
    s = _i64(sentences)
    B, L = s.shape
    i_hi = R if i_hi is None else i_hi
    u = _f32(uniforms)
    du = None if dropout_uniforms is None else _f32(dropout_uniforms)
    comp = np.zeros((R, L, B, L), np.int64)
    raw = np.zeros((R * L, B), np.float32)
    out = np.empty((B, L), np.float32) if reduce else None
    rc = lib().orc_get_reward(C.byref(gen.w), C.byref(dis.w), _p(s), B, L, R, i_lo, i_hi, col, _p(u),
                              C.c_float(dropout_p), _p(du), _p(comp), _p(raw), _p(out))
    if rc:
        raise RuntimeError("oracle: sequence length %d shorter than a filter width" % L)
    return {"completions": comp, "raw": raw, "reward": out}


def leak_rollout(gen, dis, sentences, R, uniforms, i_lo=0, i_hi=None, forced=None):
    """ROLLOUT.rollout_mc_search_leakgan for every (rollout, t) of get_reward_leakgan (utils/rollout.py:57-126,172-175).
    uniforms [R,T,L,B]; returns completions [R,T,B,L] (T = L // step_size, given_num = t*step_size + 1)."""
    s = _i64(sentences)
    B, L = s.shape
    T = L // gen.step_size
    i_hi = R if i_hi is None else i_hi
    u = _f32(uniforms)
    assert u.size == R * T * L * B
    comp = np.zeros((R, T, B, L), np.int64)
    f = None if forced is None else _i64(forced)
    rc = lib().orc_leak_rollout(C.byref(gen.w), C.byref(dis.w), _p(s), B, L, R, i_lo, i_hi, _p(u), _p(f), _p(comp))
    if rc:
        raise RuntimeError("oracle: leak_rollout rejected its arguments (%d)" % rc)
    return comp


def leak_logits_at(gen, dis, tokens, b, pos):
    """Outputs of LeakGAN_G.forward (temperature-scaled, before log_softmax) after feeding tokens[0..pos] of a row
    whose goal_init row is b: the distribution position pos+1 is drawn from."""
    t = _i64(tokens)
    out = np.empty(gen.V, np.float32)
    rc = lib().orc_leak_logits_at(C.byref(gen.w), C.byref(dis.w), _p(t), int(b), t.shape[0], int(pos), _p(out))
    if rc:
        raise RuntimeError("oracle: leak_logits_at rejected its arguments (%d)" % rc)
    return out


def get_reward_leakgan(gen, dis, sentences, R, uniforms, col=1, dropout_p=0.0, dropout_uniforms=None,
                       i_lo=0, i_hi=None, reduce=True):
    """ROLLOUT.get_reward_leakgan (utils/rollout.py:156-184).  Returns dict(completions [R,T,B,L], raw [R*T,B],
    reward [B,T])."""
    s = _i64(sentences)
    B, L = s.shape
    T = L // gen.step_size
    i_hi = R if i_hi is None else i_hi
    u = _f32(uniforms)
    assert u.size == R * T * L * B
    du = None if dropout_uniforms is None else _f32(dropout_uniforms)
    comp = np.zeros((R, T, B, L), np.int64)
    raw = np.zeros((R * T, B), np.float32)
    out = np.empty((B, T), np.float32) if reduce else None
    rc = lib().orc_get_reward_leakgan(C.byref(gen.w), C.byref(dis.w), _p(s), B, L, R, i_lo, i_hi, col, _p(u),
                                      C.c_float(dropout_p), _p(du), _p(comp), _p(raw), _p(out))
    if rc:
        raise RuntimeError("oracle: get_reward_leakgan rejected its arguments (%d)" % rc)
    return {"completions": comp, "raw": raw, "reward": out}


def mali_reward(dis, samples, R, dropout_p=0.0, dropout_uniforms=None):
    """MaliGANInstructor.get_mali_reward (instructor/real_data/maligan_instructor.py:142-154): R forward passes of D
    over the same samples (dropout mask r from dropout_uniforms[r]), mean, p/(1-p), normalise, centre, expand."""
    samples = _i64(samples)
    B, L = samples.shape
    rows = []
    for r in range(R):
        du = None if dropout_uniforms is None else dropout_uniforms[r]
        rows.append(dis_forward(dis, samples, col=1, dropout_p=dropout_p, dropout_uniforms=du)["reward"])
    s = np.zeros(B, np.float32)
    for row in rows:                       # torch.mean over the stacked dim: sequential fp32 sum / R
        s = (s + row).astype(np.float32)
    rew = (s / np.float32(R)).astype(np.float32)
    rew = (rew / (np.float32(1) - rew)).astype(np.float32)
    rew = (rew / np.sum(rew, dtype=np.float32)).astype(np.float32)
    rew = (rew - np.mean(rew, dtype=np.float32)).astype(np.float32)
    return np.broadcast_to(rew[:, None], (B, L)).copy()


def pg_loss_grads(gen, inp, target, weight, use_prob=False, padding_idx=0):
    """Policy-gradient loss and its parameter gradients, restated by hand in numpy (float64 arithmetic on the fp32
    parameters): SeqGAN_G.batchPGLoss models/SeqGAN_G.py:21-39 (use_prob False, weight = reward),
    MaliGAN_G.adv_loss models/MaliGAN_G.py:21-39 (same), SentiGAN_G.batchPGLoss models/SentiGAN_G.py:47-65
    (use_prob True, weight = 1 - reward).  loss = -sum(pred * weight), pred = log-softmax (or softmax) of
    lstm2out(lstm(emb(inp))) picked at the target; the gradients are what autograd returns for the seven
    parameters of LSTMGenerator (models/generator.py:33-35), the embedding's padding_idx row left at zero.
    Returns (loss, dict of gradients keyed like the state_dict)."""
    emb, w_ih, w_hh, b_ih, b_hh, w_out, b_out = [x.astype(np.float64) for x in gen.a]
    inp, target = _i64(inp), _i64(target)
    N, L = inp.shape
    H, V = gen.H, gen.V
    weight = np.broadcast_to(np.asarray(weight, np.float64), (N, L))
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    h = np.zeros((N, H)); c = np.zeros((N, H))
    hs, cs, gates = [h], [c], []
    for t in range(L):  # nn.LSTM, gate order i,f,g,o (models/generator.py:51)
        z = emb[inp[:, t]] @ w_ih.T + h @ w_hh.T + b_ih + b_hh
        i, f, g, o = sig(z[:, :H]), sig(z[:, H:2 * H]), np.tanh(z[:, 2 * H:3 * H]), sig(z[:, 3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        hs.append(h); cs.append(c); gates.append((i, f, g, o))
    hall = np.stack(hs[1:], 1)                                  # [N,L,H]
    logits = hall @ w_out.T + b_out                             # lstm2out (models/generator.py:53)
    m = logits.max(-1, keepdims=True)
    logsm = logits - m - np.log(np.exp(logits - m).sum(-1, keepdims=True))
    logp = np.take_along_axis(logsm, target[..., None], -1)[..., 0]
    pred = np.exp(logp) if use_prob else logp
    loss = -(pred * weight).sum()
    coeff = -weight * (np.exp(logp) if use_prob else 1.0)       # dLoss/dlogp
    dlogits = -np.exp(logsm) * coeff[..., None]
    np.add.at(dlogits, (np.arange(N)[:, None], np.arange(L)[None, :], target), coeff)
    d = {"lstm2out.weight": np.einsum("ntv,nth->vh", dlogits, hall), "lstm2out.bias": dlogits.sum((0, 1))}
    dh_proj = dlogits @ w_out                                   # [N,L,H]
    d_emb = np.zeros_like(emb); d_wih = np.zeros_like(w_ih); d_whh = np.zeros_like(w_hh); d_b = np.zeros(4 * H)
    dh_next = np.zeros((N, H)); dc_next = np.zeros((N, H))
    for t in range(L - 1, -1, -1):
        i, f, g, o = gates[t]
        tc = np.tanh(cs[t + 1])
        dh = dh_proj[:, t] + dh_next
        dc = dc_next + dh * o * (1 - tc * tc)
        da = np.concatenate([dc * g * i * (1 - i), dc * cs[t] * f * (1 - f), dc * i * (1 - g * g),
                             dh * tc * o * (1 - o)], 1)         # [N,4H] pre-activation gradients
        x = emb[inp[:, t]]
        d_wih += da.T @ x; d_whh += da.T @ hs[t]; d_b += da.sum(0)
        np.add.at(d_emb, inp[:, t], da @ w_ih)
        dh_next = da @ w_hh
        dc_next = dc * f
    if padding_idx is not None and 0 <= padding_idx < V:
        d_emb[padding_idx] = 0.0
    d.update({"embeddings.weight": d_emb, "lstm.weight_ih_l0": d_wih, "lstm.weight_hh_l0": d_whh,
              "lstm.bias_ih_l0": d_b, "lstm.bias_hh_l0": d_b.copy()})
    return float(loss), {k: v.astype(np.float32) for k, v in d.items()}
