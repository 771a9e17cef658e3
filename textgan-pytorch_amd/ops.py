"""Tensor-level wrappers of the C ABI (one function per entry point).

All tensors are CUDA (ROCm) tensors; work is enqueued on torch's current
stream; outputs are allocated by torch and written by the HIP kernels.
"""
import ctypes as C
import functools

import torch

from . import _lib

I64, F32 = torch.int64, torch.float32


def _on_device(fn):
    """The C ABI launches on the CURRENT HIP device and stream: make the device of the call's tensors / modules
    current for its duration (a rank that holds its tensors on cuda:k without torch.cuda.set_device(k) would
    otherwise launch on device 0's stream against device-k pointers), and refuse a call that mixes devices."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        dev = None
        for a in list(args) + list(kw.values()):
            d = None
            if isinstance(a, torch.Tensor):
                d = a.device
            elif isinstance(a, torch.nn.Module):
                d = next((p.device for p in a.parameters()), None)
            if d is None or d.type != "cuda":   # CPU tensors are diagnosed where their pointer is taken
                continue
            if dev is None:
                dev = d
            elif d != dev:
                raise RuntimeError("textgan_hip: the tensors and modules of one call must live on one device "
                                   "(got %s and %s)" % (dev, d))
        if dev is None:
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return wrapped


def _tokens(t, name):
    if t.dtype != I64:
        t = t.long()
    return t.contiguous() if not t.is_contiguous() else t, name


@_on_device
def gen_teacher_forced(gen, inp, target=None, want_logp=True, want_lse=False, want_logits=False, want_states=False,
                       states_in=None):
    """tg_gen_teacher_forced_ws: LSTMGenerator.forward over [N,L] (models/generator.py:40-60) from the zero state or,
    with ``states_in`` [N,2,H], from a carried (h, c).  Returns dict(logp [N,L], lse [N,L], logits [N,L,V],
    states [L+1,N,2,H])."""
    w, keep = _lib.gen_weights(gen, teacher=True)
    inp, _ = _tokens(inp, "inp")
    N, L = inp.shape
    dev = inp.device
    tgt = None if target is None else _tokens(target, "target")[0]
    logp = torch.empty(N, L, dtype=F32, device=dev) if (want_logp and tgt is not None) else None
    lse = torch.empty(N, L, dtype=F32, device=dev) if want_lse else None
    logits = torch.empty(N, L, w.V, dtype=F32, device=dev) if want_logits else None
    states = torch.empty(L + 1, N, 2, w.H, dtype=F32, device=dev) if want_states else None
    if states_in is not None:
        if tuple(states_in.shape) != (N, 2, w.H):
            raise RuntimeError("textgan_hip: states_in must be [N, 2, H] = [%d, 2, %d], got %s"
                               % (N, w.H, tuple(states_in.shape)))
        states_in = states_in.float().contiguous()
    # shapes outside the persistent kernel's set run as fp32 GEMMs and need scratch (0 bytes otherwise)
    nbytes = _lib.lib().tg_gen_teacher_forced_workspace_bytes(C.byref(w), N, L, int(want_logits), int(want_states))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    _lib.check(_lib.lib().tg_gen_teacher_forced_ws(
        C.byref(w), _lib.dev_ptr(inp, I64, "inp"), _lib.dev_ptr(tgt, I64, "target"), N, L,
        _lib.dev_ptr(states_in, F32, "states_in"), _lib.dev_ptr(logp), _lib.dev_ptr(lse), _lib.dev_ptr(logits),
        _lib.dev_ptr(states), _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return {"logp": logp, "lse": lse, "logits": logits, "states": states}


@_on_device
def gen_nll(gen, inp, target, precision=None, want_lse=False):
    """tg_gen_nll: log-probabilities of the targets under teacher forcing (metrics/nll.py:59-61), in the
    library's default generator arithmetic or ``precision``.  Returns logp [N,L] (and lse if asked)."""
    w, keep = _lib.gen_weights(gen, precision)
    inp, _ = _tokens(inp, "inp")
    target, _ = _tokens(target, "target")
    N, L = inp.shape
    dev = inp.device
    logp = torch.empty(N, L, dtype=F32, device=dev)
    lse = torch.empty(N, L, dtype=F32, device=dev) if want_lse else None
    nbytes = _lib.lib().tg_gen_nll_workspace_bytes(C.byref(w), N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    _lib.check(_lib.lib().tg_gen_nll(
        C.byref(w), _lib.dev_ptr(inp, I64, "inp"), _lib.dev_ptr(target, I64, "target"), N, L,
        _lib.dev_ptr(logp), _lib.dev_ptr(lse), _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return (logp, lse) if want_lse else logp


@_on_device
def gen_pg_forward(gen, inp, target, precision=None):
    """tg_gen_pg_forward: logp / lse / states of a teacher-forced pass, row-parallel projection (few sequences).
    Same dict as gen_teacher_forced(want_lse=True, want_states=True).  precision (beyond E = H = 32): arithmetic of the
    vocabulary product, "split_f16" (default) or "fp32"."""
    w, keep = _lib.gen_weights(gen, precision, teacher=True)
    inp, _ = _tokens(inp, "inp")
    target, _ = _tokens(target, "target")
    N, L = inp.shape
    dev = inp.device
    logp = torch.empty(N, L, dtype=F32, device=dev)
    lse = torch.empty(N, L, dtype=F32, device=dev)
    states = torch.empty(L + 1, N, 2, w.H, dtype=F32, device=dev)
    nbytes = _lib.lib().tg_gen_pg_forward_workspace_bytes(C.byref(w), N, L)   # 0 for E = H = 32
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    _lib.check(_lib.lib().tg_gen_pg_forward(
        C.byref(w), _lib.dev_ptr(inp, I64, "inp"), _lib.dev_ptr(target, I64, "target"), N, L,
        _lib.dev_ptr(logp), _lib.dev_ptr(lse), _lib.dev_ptr(states), _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return {"logp": logp, "lse": lse, "logits": None, "states": states}


@_on_device
def gen_pg_backward(gen, inp, target, coeff, lse, states, padding_idx=-1, precision=None):
    """tg_gen_pg_backward: gradients of sum(coeff-weighted log-probabilities of the targets) with respect to the
    seven generator parameters (what autograd produces behind SeqGAN_G.batchPGLoss, models/SeqGAN_G.py:21-39).
    Returns a tuple in GenW order: (emb, w_ih, w_hh, b_ih, b_hh, w_out, b_out).  precision: as gen_pg_forward (the two
    halves of one step must use the same one: the backward recomputes the logits the forward's lse belongs to)."""
    w, keep = _lib.gen_weights(gen, precision, teacher=True)
    inp, _ = _tokens(inp, "inp")
    target, _ = _tokens(target, "target")
    N, L = inp.shape
    dev = inp.device
    shapes = [(w.V, w.E), (4 * w.H, w.E), (4 * w.H, w.H), (4 * w.H,), (4 * w.H,), (w.V, w.H), (w.V,)]
    grads = [torch.empty(s, dtype=F32, device=dev) for s in shapes]
    g = _lib.GenGrads(*[t.data_ptr() for t in grads])
    nbytes = _lib.lib().tg_gen_pg_backward_workspace_bytes(C.byref(w), N, L)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    coeff = coeff.contiguous()
    _lib.check(_lib.lib().tg_gen_pg_backward(
        C.byref(w), _lib.dev_ptr(inp, I64, "inp"), _lib.dev_ptr(target, I64, "target"),
        _lib.dev_ptr(coeff, F32, "coeff"), _lib.dev_ptr(lse, F32, "lse"), _lib.dev_ptr(states, F32, "states"),
        N, L, int(padding_idx), C.byref(g), _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return tuple(grads)


@_on_device
def gen_rollout(gen, sentences, R, uniforms, forced=None, g_lo=1, g_hi=None, i_lo=0, i_hi=None, out=None,
                precision=None):
    """tg_gen_rollout: every rollout_mc_search(sentences, g) of get_reward's loops
    (utils/rollout.py:25-55,143-145).  uniforms [R,L,L,B]; returns completions [R,L,B,L] int64."""
    w, keep = _lib.gen_weights(gen, precision)
    s, _ = _tokens(sentences, "sentences")
    B, L = s.shape
    g_hi = L if g_hi is None else g_hi
    i_hi = R if i_hi is None else i_hi
    if uniforms.numel() != R * L * L * B:
        raise RuntimeError("textgan_hip: uniforms must have R*L*L*B = %d elements, got %d"
                           % (R * L * L * B, uniforms.numel()))
    comp = out if out is not None else torch.empty(R, L, B, L, dtype=I64, device=s.device)
    nbytes = _lib.lib().tg_gen_rollout_workspace_bytes(C.byref(w), B, L, i_hi - i_lo)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
    f = None if forced is None else _tokens(forced, "forced")[0]
    _lib.check(_lib.lib().tg_gen_rollout(
        C.byref(w), _lib.dev_ptr(s, I64, "sentences"), B, L, R, g_lo, g_hi, i_lo, i_hi,
        _lib.dev_ptr(uniforms, F32, "uniforms"), _lib.dev_ptr(f, I64, "forced"), _lib.dev_ptr(comp, I64, "completions"),
        _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return comp


@_on_device
def leak_rollout(gen, dis, sentences, R, given_nums, uniforms, forced=None, i_lo=0, i_hi=None, out=None,
                 dis_weights=None, dis_precision=None, precision=None):
    """tg_leak_rollout: every rollout_mc_search_leakgan(sentences, dis, given_num) of get_reward_leakgan's loops
    (utils/rollout.py:57-126,172-175).  given_nums: the T given_num values (t*step_size + 1 for get_reward_leakgan);
    uniforms [R,T,L,B]; returns completions [R,T,B,L] int64."""
    w, keep = _lib.leak_weights(gen, precision)
    dw, dkeep = dis_weights if dis_weights is not None else _lib.dis_weights(dis, dis_precision)
    s, _ = _tokens(sentences, "sentences")
    B, L = s.shape
    T = len(given_nums)
    i_hi = R if i_hi is None else i_hi
    if uniforms.numel() != R * T * L * B:
        raise RuntimeError("textgan_hip: uniforms must have R*T*L*B = %d elements, got %d"
                           % (R * T * L * B, uniforms.numel()))
    comp = out if out is not None else torch.empty(R, T, B, L, dtype=I64, device=s.device)
    given = (C.c_int32 * T)(*[int(g) for g in given_nums])
    nbytes = _lib.lib().tg_leak_rollout_workspace_bytes(C.byref(w), C.byref(dw), B, L, T, max(1, i_hi - i_lo))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=s.device)
    f = None if forced is None else _tokens(forced, "forced")[0]
    _lib.check(_lib.lib().tg_leak_rollout(
        C.byref(w), C.byref(dw), _lib.dev_ptr(s, I64, "sentences"), B, L, R, given, T, i_lo, i_hi,
        _lib.dev_ptr(uniforms, F32, "uniforms"), _lib.dev_ptr(f, I64, "forced"), _lib.dev_ptr(comp, I64, "completions"),
        _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return comp


@_on_device
def gen_sample(gen, N, L, start_letter, uniforms, forced=None, precision=None):
    """tg_gen_sample: the step loop of LSTMGenerator.sample (models/generator.py:71-81) for N rows.
    uniforms [L,N]; returns [N,L] int64 on the device."""
    w, keep = _lib.gen_weights(gen, precision)
    if uniforms.numel() != L * N:
        raise RuntimeError("textgan_hip: uniforms must have L*N = %d elements" % (L * N))
    out = torch.empty(N, L, dtype=I64, device=uniforms.device)
    f = None if forced is None else _tokens(forced, "forced")[0]
    nbytes = _lib.lib().tg_gen_sample_workspace_bytes(C.byref(w), N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=uniforms.device) if nbytes else None
    _lib.check(_lib.lib().tg_gen_sample(C.byref(w), N, L, int(start_letter), _lib.dev_ptr(uniforms, F32, "uniforms"),
                                        _lib.dev_ptr(f, I64, "forced"), _lib.dev_ptr(out), _lib.dev_ptr(ws), nbytes,
                                        _lib.stream_ptr()))
    return out


def _dropout_args(w, N, dropout_p, dropout_uniforms, dropout_seed, count):
    """Explicit uniforms (parity / reproducibility) or, without them, a seed for the in-kernel counter-based masks."""
    if dropout_p > 0.0 and dropout_uniforms is not None:
        if dropout_uniforms.numel() != count:
            raise RuntimeError("textgan_hip: dropout_uniforms must have %d elements, got %d"
                               % (count, dropout_uniforms.numel()))
        return _lib.dev_ptr(dropout_uniforms, F32, "dropout_uniforms")
    if dropout_p > 0.0:
        w.dropout_seed = _lib.next_dropout_seed() if dropout_seed is None else int(dropout_seed) & 0xffffffff
    return None


@_on_device
def dis_score_prepare(dis, N, L, device, dropout_uniforms=None, weights=None, precision=None):
    """tg_dis_score_prepare: the part of the scorer that needs only D's parameters, enqueued on the current stream.
    Returns the workspace to hand to dis_score(..., prepared=ws) for N sequences of length L (rewards only)."""
    w, keep = weights if weights is not None else _lib.dis_weights(dis, precision)
    nbytes = _lib.lib().tg_dis_workspace_bytes(C.byref(w), N, L)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
    du = None if dropout_uniforms is None else _lib.dev_ptr(dropout_uniforms, F32, "dropout_uniforms")
    _lib.check(_lib.lib().tg_dis_score_prepare(C.byref(w), N, L, du, None, _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return ws


@_on_device
def dis_score(dis, seqs, col=1, dropout_p=0.0, dropout_uniforms=None, want_reward=True, want_logits=False,
              want_feature=False, reward_out=None, weights=None, precision=None, dropout_seed=None, prepared=None):
    """tg_dis_score: CNNDiscriminator.forward (models/discriminator.py:40-64) + softmax column
    (utils/rollout.py:146-148).  seqs [N,L]; returns dict(reward [N], logits [N,n_out], feature [N,F]).
    prepared: the workspace dis_score_prepare returned for the same D, N and L (no features then: the preparation
    was made for the pointers of a rewards-only call)."""
    w, keep = weights if weights is not None else _lib.dis_weights(dis, precision)
    s, _ = _tokens(seqs, "seqs")
    N, L = s.shape
    dev = s.device
    reward = reward_out if reward_out is not None else (torch.empty(N, dtype=F32, device=dev) if want_reward else None)
    logits = torch.empty(N, w.n_out, dtype=F32, device=dev) if want_logits else None
    feat = torch.empty(N, w.F, dtype=F32, device=dev) if want_feature else None
    du = _dropout_args(w, N, dropout_p, dropout_uniforms, dropout_seed, N * w.F)
    nbytes = _lib.lib().tg_dis_workspace_bytes(C.byref(w), N, L)
    if prepared is not None and (prepared.numel() < nbytes or want_feature):
        raise RuntimeError("textgan_hip: the prepared workspace does not fit this call")
    ws = prepared if prepared is not None else torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    fn = _lib.lib().tg_dis_score_prepared if prepared is not None else _lib.lib().tg_dis_score
    _lib.check(fn(
        C.byref(w), _lib.dev_ptr(s, I64, "seqs"), N, L, int(col), float(dropout_p),
        du, _lib.dev_ptr(reward), _lib.dev_ptr(logits),
        _lib.dev_ptr(feat), _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return {"reward": reward, "logits": logits, "feature": feat}


@_on_device
def dis_score_rollout_prepare(dis, sentences, R, dropout_uniforms=None, weights=None, precision=None):
    """tg_dis_score_rollout_prepare: the part of the rollout scorer that needs only D's parameters and the sentences,
    enqueued on the current stream.  Returns the workspace to hand to dis_score_rollout(..., prepared=ws); the caller
    orders the two calls (ROLLOUT.get_reward runs this one on a second stream beside the generator)."""
    w, keep = weights if weights is not None else _lib.dis_weights(dis, precision)
    s, _ = _tokens(sentences, "sentences")
    B, L = s.shape
    nbytes = _lib.lib().tg_dis_score_rollout_workspace_bytes(C.byref(w), B, L, R)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=s.device)
    du = None if dropout_uniforms is None else _lib.dev_ptr(dropout_uniforms, F32, "dropout_uniforms")
    _lib.check(_lib.lib().tg_dis_score_rollout_prepare(
        C.byref(w), _lib.dev_ptr(s, I64, "sentences"), B, L, R, du, _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return ws


@_on_device
def dis_score_rollout(dis, sentences, completions, col=1, dropout_p=0.0, dropout_uniforms=None, reward_out=None,
                      weights=None, precision=None, dropout_seed=None, prepared=None):
    """tg_dis_score_rollout: the scorer over tg_gen_rollout's [R,L,B,L] completions with prefix reuse
    (windows inside the tokens a completion shares with its sentence are evaluated once per sentence).
    Returns raw reward rows [R*L, B].  prepared: the workspace dis_score_rollout_prepare returned for the same D,
    sentences and R (tg_dis_score_rollout_prepared then runs the rest)."""
    w, keep = weights if weights is not None else _lib.dis_weights(dis, precision)
    s, _ = _tokens(sentences, "sentences")
    comp, _ = _tokens(completions, "completions")
    R, L, B, L2 = comp.shape
    if tuple(s.shape) != (B, L) or L2 != L:
        raise RuntimeError("textgan_hip: completions must be [R,L,B,L] matching sentences [B,L]")
    dev = s.device
    reward = reward_out if reward_out is not None else torch.empty(R * L, B, dtype=F32, device=dev)
    du = _dropout_args(w, R * L * B, dropout_p, dropout_uniforms, dropout_seed, R * L * B * w.F)
    nbytes = _lib.lib().tg_dis_score_rollout_workspace_bytes(C.byref(w), B, L, R)
    if prepared is not None and prepared.numel() < nbytes:
        raise RuntimeError("textgan_hip: the prepared workspace holds %d bytes, this call needs %d"
                           % (prepared.numel(), nbytes))
    ws = prepared if prepared is not None else torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    fn = _lib.lib().tg_dis_score_rollout_prepared if prepared is not None else _lib.lib().tg_dis_score_rollout
    _lib.check(fn(
        C.byref(w), _lib.dev_ptr(s, I64, "sentences"), _lib.dev_ptr(comp, I64, "completions"), B, L, R, int(col),
        float(dropout_p), du, _lib.dev_ptr(reward, F32, "reward"),
        _lib.dev_ptr(ws), nbytes, _lib.stream_ptr()))
    return reward


@_on_device
def reward_reduce(raw, B, L, R):
    """tg_reward_reduce: torch.mean(rewards.view(B, L, R), dim=-1) (utils/rollout.py:153)."""
    if raw.numel() != R * L * B:
        raise RuntimeError("textgan_hip: raw must have R*L*B elements")
    out = torch.empty(B, L, dtype=F32, device=raw.device)
    _lib.check(_lib.lib().tg_reward_reduce(_lib.dev_ptr(raw, F32, "raw"), B, L, R, _lib.dev_ptr(out),
                                           _lib.stream_ptr()))
    return out


@_on_device
def dis_rescore_dropout(dis, feature, R, col=1, dropout_p=0.0, dropout_uniforms=None, weights=None,
                        dropout_seed=None):
    """tg_dis_rescore_dropout: R dropout masks + output layer + softmax column over precomputed highway features
    (the loop of MaliGANInstructor.get_mali_reward, instructor/real_data/maligan_instructor.py:142-147).
    feature [N,F]; dropout_uniforms [R,N,F]; returns [R,N]."""
    w, keep = weights if weights is not None else _lib.dis_weights(dis)
    N = feature.shape[0]
    if feature.dim() != 2 or feature.shape[1] != w.F:
        raise RuntimeError("textgan_hip: feature must be [N, %d]" % w.F)
    du = _dropout_args(w, R * N, dropout_p, dropout_uniforms, dropout_seed, R * N * w.F)
    out = torch.empty(R, N, dtype=F32, device=feature.device)
    _lib.check(_lib.lib().tg_dis_rescore_dropout(
        C.byref(w), _lib.dev_ptr(feature, F32, "feature"), N, R, col, float(dropout_p), du,
        _lib.dev_ptr(out), _lib.stream_ptr()))
    return out
