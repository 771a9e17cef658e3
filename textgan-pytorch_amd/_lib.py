"""ctypes binding of libtextgan_hip.so (the C ABI in include/textgan_hip.h).

There is no CPU fallback: if the library is missing or a tensor is not a CUDA
(ROCm) tensor, every entry point raises RuntimeError.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TEXTGAN_HIP_LIB: another build of the same library (diagnostic builds with in-kernel stamps, tools/ubench)
LIB_PATH = os.environ.get("TEXTGAN_HIP_LIB") or os.path.join(_HERE, "libtextgan_hip.so")
MAX_FILTERS = 16
MAX_OUT = 8
ABI_VERSION = b"textgan_hip 0.4"   # include/textgan_hip.h; argument lists / structs changed in 0.2, 0.3, 0.4

SYMBOLS = ["tg_version", "tg_last_error", "tg_gen_supported", "tg_gen_teacher_forced",
           "tg_gen_teacher_forced_workspace_bytes", "tg_gen_teacher_forced_ws", "tg_gen_pg_forward_workspace_bytes",
           "tg_gen_rollout_workspace_bytes", "tg_gen_rollout", "tg_gen_sample_workspace_bytes", "tg_gen_sample",
           "tg_gen_sample_static_bytes", "tg_gen_sample_static",
           "tg_dis_workspace_bytes",
           "tg_dis_score", "tg_dis_score_prepare", "tg_dis_score_prepared", "tg_dis_score_rollout_workspace_bytes", "tg_dis_score_rollout", "tg_dis_score_rollout_prepare",
           "tg_dis_score_rollout_prepared", "tg_dis_rescore_dropout",
           "tg_gen_nll_workspace_bytes", "tg_gen_nll", "tg_gen_pg_forward", "tg_gen_pg_backward_workspace_bytes", "tg_gen_pg_backward",
           "tg_leak_rollout_workspace_bytes", "tg_leak_rollout",
           "tg_reward_reduce", "tg_allgather_rows",
           "tg_profile_enable", "tg_profile_read"]

KERNELS = {"gen_teacher": 0, "gen_sample": 1, "dis_conv": 2, "dis_highway": 3, "dis_final": 4, "reduce": 5,
           "dis_prefix": 6, "pg_backward": 7, "pg_forward": 8, "genw_cell": 9, "genw_proj": 10, "genw_draw": 11,
           "leak_draw": 12, "leak_cells": 13, "leak_proj": 14, "pgw_forward": 15, "pgw_backward": 16, "pgw_proj": 17, "pgw_proj_dh": 18,
           "pgw_proj_dw": 19}


class GenW(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out")] + \
               [("V", C.c_int32), ("E", C.c_int32), ("H", C.c_int32), ("precision", C.c_int32)]


class GenGrads(C.Structure):
    """TgGenGrads: gradient buffers, same order and shapes as the GenW pointers."""
    _fields_ = [(k, C.c_void_p) for k in ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out")]


class LeakW(C.Structure):
    """TgLeakWeights: LeakGAN_G parameters (models/LeakGAN_G.py:40-48)."""
    _fields_ = [(k, C.c_void_p) for k in ("emb", "w_wih", "w_whh", "w_bih", "w_bhh", "m_wih", "m_whh", "m_bih", "m_bhh",
                                          "w2g_w", "w2g_b", "m2g_w", "m2g_b", "g2g_w", "goal_init")] + \
               [(k, C.c_int32) for k in ("V", "E", "H", "G", "K", "step_size", "B_init")] + [("temperature", C.c_float),
                                                                                          ("precision", C.c_int32)]


class DisW(C.Structure):
    _fields_ = [("emb", C.c_void_p), ("conv_w", C.c_void_p * MAX_FILTERS), ("conv_b", C.c_void_p * MAX_FILTERS),
                ("hw_w", C.c_void_p), ("hw_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
                ("f", C.c_int32 * MAX_FILTERS), ("n", C.c_int32 * MAX_FILTERS), ("n_filters", C.c_int32),
                ("V", C.c_int32), ("Ed", C.c_int32), ("F", C.c_int32), ("n_out", C.c_int32), ("precision", C.c_int32),
                ("dropout_seed", C.c_uint32)]


_lib = None


def lib():
    """Load the HIP library (once). Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("textgan HIP extension not built: %s is missing "
                               "(run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C textgan-pytorch_amd/csrc`)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.tg_version.restype = C.c_char_p
        if ABI_VERSION not in l.tg_version():
            raise RuntimeError("textgan_hip: %s is %r, this binding needs ABI %s -- rebuild it (make -C "
                               "textgan-pytorch_amd/csrc)" % (LIB_PATH, l.tg_version(), ABI_VERSION.decode()))
        l.tg_last_error.restype = C.c_char_p
        l.tg_gen_rollout_workspace_bytes.restype = C.c_size_t
        l.tg_gen_rollout_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        l.tg_gen_sample_workspace_bytes.restype = C.c_size_t
        l.tg_gen_sample_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
        l.tg_dis_workspace_bytes.restype = C.c_size_t
        l.tg_dis_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.tg_gen_teacher_forced.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 5
        l.tg_gen_rollout.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 4 + [C.c_size_t,
                                                                                                  C.c_void_p]
        l.tg_gen_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]
        l.tg_gen_sample_static_bytes.restype = C.c_size_t
        l.tg_gen_sample_static_bytes.argtypes = []
        l.tg_gen_sample_static.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
        l.tg_gen_sample_static.restype = C.c_int
        l.tg_dis_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 5 + \
                                  [C.c_size_t, C.c_void_p]
        l.tg_dis_score_prepared.argtypes = l.tg_dis_score.argtypes
        l.tg_dis_score_prepared.restype = C.c_int
        l.tg_dis_score_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p]
        l.tg_dis_score_prepare.restype = C.c_int
        l.tg_dis_score_rollout_workspace_bytes.restype = C.c_size_t
        l.tg_dis_score_rollout_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        l.tg_dis_score_rollout.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float] + [C.c_void_p] * 3 + \
                                          [C.c_size_t, C.c_void_p]
        l.tg_dis_score_rollout.restype = C.c_int
        l.tg_dis_score_rollout_prepared.argtypes = l.tg_dis_score_rollout.argtypes
        l.tg_dis_score_rollout_prepared.restype = C.c_int
        l.tg_dis_score_rollout_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_size_t, C.c_void_p]
        l.tg_dis_score_rollout_prepare.restype = C.c_int
        l.tg_dis_rescore_dropout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        l.tg_dis_rescore_dropout.restype = C.c_int
        l.tg_reward_reduce.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.tg_gen_pg_backward_workspace_bytes.restype = C.c_size_t
        l.tg_gen_pg_backward_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.tg_gen_pg_backward.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                                           C.c_size_t, C.c_void_p]
        l.tg_gen_pg_backward.restype = C.c_int
        l.tg_gen_nll_workspace_bytes.restype = C.c_size_t
        l.tg_gen_nll_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
        l.tg_gen_nll.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p]
        l.tg_gen_nll.restype = C.c_int
        l.tg_gen_pg_forward.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
        l.tg_gen_pg_forward_workspace_bytes.restype = C.c_size_t
        l.tg_gen_pg_forward_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.tg_gen_teacher_forced_workspace_bytes.restype = C.c_size_t
        l.tg_gen_teacher_forced_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        l.tg_gen_teacher_forced_ws.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]
        l.tg_gen_teacher_forced_ws.restype = C.c_int
        l.tg_gen_pg_forward.restype = C.c_int
        l.tg_leak_rollout_workspace_bytes.restype = C.c_size_t
        l.tg_leak_rollout_workspace_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        l.tg_leak_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p]
        l.tg_leak_rollout.restype = C.c_int
        l.tg_allgather_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        l.tg_allgather_rows.restype = C.c_int
        l.tg_profile_enable.argtypes = [C.c_int]
        l.tg_profile_enable.restype = None
        l.tg_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        l.tg_profile_read.restype = C.c_int
        for fn in ("tg_gen_supported", "tg_gen_teacher_forced", "tg_gen_rollout", "tg_gen_sample", "tg_dis_score",
                   "tg_reward_reduce"):
            getattr(l, fn).restype = C.c_int
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("textgan_hip: %s (code %d)" % (lib().tg_last_error().decode(), rc))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("textgan_hip: %s must be a CUDA (ROCm) tensor, got device %s -- there is no CPU path"
                           % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("textgan_hip: %s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("textgan_hip: %s must be contiguous" % name)
    return t.data_ptr()


def _param(p, name):
    d = p.detach()
    if d.dtype != torch.float32 or not d.is_contiguous():
        d = d.float().contiguous()
    return d, dev_ptr(d, torch.float32, name)


PRECISIONS = {"fp32": 0, "split_f16": 1}
#: arithmetic used when a caller does not say: see TG_GEN_* / TG_DIS_* in include/textgan_hip.h
default_gen_precision = "split_f16"
default_dis_precision = "split_f16"


def gen_weights(gen, precision=None, teacher=False):
    """Pack the live parameters of an LSTMGenerator-shaped module (models/generator.py:33-35).
    Read on every call: the generator is stepped between get_reward calls.  teacher: the call is a teacher-forced /
    policy-gradient pass, which every shape of tg_gen_supported has in fp32 (the samplers need split_f16 beyond the
    persistent kernel's set)."""
    keep = []
    ptrs = []
    for name, p in (("embeddings.weight", gen.embeddings.weight), ("lstm.weight_ih_l0", gen.lstm.weight_ih_l0),
                    ("lstm.weight_hh_l0", gen.lstm.weight_hh_l0), ("lstm.bias_ih_l0", gen.lstm.bias_ih_l0),
                    ("lstm.bias_hh_l0", gen.lstm.bias_hh_l0), ("lstm2out.weight", gen.lstm2out.weight),
                    ("lstm2out.bias", gen.lstm2out.bias)):
        d, ptr = _param(p, name)
        keep.append(d)
        ptrs.append(ptr)
    V, E = keep[0].shape
    H = keep[2].shape[1]
    if keep[1].shape != (4 * H, E) or keep[5].shape != (V, H):
        raise RuntimeError("textgan_hip: generator parameter shapes are inconsistent")
    w = GenW(*ptrs, V, E, H, PRECISIONS[precision or default_gen_precision])
    sup = lib().tg_gen_supported(V, E, H)
    if sup < 0 or (sup == 1 and not teacher and (precision or default_gen_precision) != "split_f16"):
        raise RuntimeError("textgan_hip: generator shape V=%d E=%d H=%d is not supported by the compiled kernels "
                           "(every entry point: H in {32,64,128,256,512}, E a multiple of 32 up to 512; rollout / sample "
                           "/ NLL in split_f16: E, H multiples of 32 up to 1024)" % (V, E, H))
    return w, keep


def leak_weights(gen, precision=None):
    """Pack the live parameters of a LeakGAN_G-shaped module (models/LeakGAN_G.py:40-48): embeddings, worker and
    manager LSTMs, work2goal, mana2goal, goal2goal, goal_init, with step_size / temperature from the module.
    precision: arithmetic of the work2goal projection, "fp32" or "split_f16" (None: default_gen_precision)."""
    names = (("embeddings.weight", gen.embeddings.weight),
             ("worker.weight_ih_l0", gen.worker.weight_ih_l0), ("worker.weight_hh_l0", gen.worker.weight_hh_l0),
             ("worker.bias_ih_l0", gen.worker.bias_ih_l0), ("worker.bias_hh_l0", gen.worker.bias_hh_l0),
             ("manager.weight_ih_l0", gen.manager.weight_ih_l0), ("manager.weight_hh_l0", gen.manager.weight_hh_l0),
             ("manager.bias_ih_l0", gen.manager.bias_ih_l0), ("manager.bias_hh_l0", gen.manager.bias_hh_l0),
             ("work2goal.weight", gen.work2goal.weight), ("work2goal.bias", gen.work2goal.bias),
             ("mana2goal.weight", gen.mana2goal.weight), ("mana2goal.bias", gen.mana2goal.bias),
             ("goal2goal.weight", gen.goal2goal.weight), ("goal_init", gen.goal_init))
    keep, ptrs = [], []
    for name, p in names:
        d, ptr = _param(p, name)
        keep.append(d)
        ptrs.append(ptr)
    V, E = keep[0].shape
    H = keep[2].shape[1]
    K, G = keep[13].shape
    if (keep[1].shape != (4 * H, E) or keep[5].shape != (4 * H, G) or keep[9].shape != (V * K, H)
            or keep[11].shape != (G, H) or keep[14].shape[1] != G):
        raise RuntimeError("textgan_hip: LeakGAN generator parameter shapes are inconsistent")
    if (precision or default_gen_precision) not in PRECISIONS:
        raise RuntimeError("textgan_hip: unknown precision %r" % (precision,))
    w = LeakW(*ptrs, V, E, H, G, K, int(gen.step_size), int(keep[14].shape[0]), float(getattr(gen, "temperature", 1.5)),
              PRECISIONS[precision or default_gen_precision])
    return w, keep


DIS_EMBED_DIM = 64   # kEd of csrc/tg_dis.hip: the scorer kernels' embedding width (the reference's cfg.dis_embed_dim)


def _pad_last(d, width, name):
    """[.., Ed] -> [.., 64] with zero columns: a narrower discriminator embedding (CNNDiscriminator takes any embed_dim,
    models/discriminator.py:20-36) is the 64-wide one whose extra embedding columns and conv-weight columns are zero --
    every window product gains only exact zeros, so the results are those of the Ed-wide scorer."""
    pad = torch.zeros(d.shape[:-1] + (width,), dtype=torch.float32, device=d.device)
    pad[..., :d.shape[-1]] = d
    return pad, dev_ptr(pad, torch.float32, name)


def dis_weights(dis, precision=None):
    """Pack the live parameters of a CNNDiscriminator-shaped module (models/discriminator.py:30-36).  embed_dim 64
    (the reference's default) is read in place; a narrower one is zero-padded to 64 per call; a wider one is refused."""
    keep = []
    w = DisW()
    w.precision = PRECISIONS[precision or default_dis_precision]
    d, w.emb = _param(dis.embeddings.weight, "embeddings.weight")
    V, Ed = d.shape
    if Ed > DIS_EMBED_DIM:
        raise RuntimeError("textgan_hip: discriminator embed_dim %d is wider than the %d the scorer kernels are built for"
                           % (Ed, DIS_EMBED_DIM))
    if Ed < DIS_EMBED_DIM:
        d, w.emb = _pad_last(d, DIS_EMBED_DIM, "embeddings.weight")
    keep.append(d)
    if len(dis.convs) > MAX_FILTERS:
        raise RuntimeError("textgan_hip: more than %d filter widths" % MAX_FILTERS)
    F = 0
    for k, conv in enumerate(dis.convs):
        cw, w.conv_w[k] = _param(conv.weight, "convs.%d.weight" % k)
        cb, w.conv_b[k] = _param(conv.bias, "convs.%d.bias" % k)
        n, one, f, ed = cw.shape
        if one != 1 or ed != Ed:
            raise RuntimeError("textgan_hip: convs.%d.weight has shape %s, expected [n,1,f,%d]" % (k, tuple(cw.shape), Ed))
        if Ed < DIS_EMBED_DIM:
            cw, w.conv_w[k] = _pad_last(cw, DIS_EMBED_DIM, "convs.%d.weight" % k)
        keep += [cw, cb]
        w.f[k], w.n[k] = f, n
        F += n
    w.n_filters = len(dis.convs)
    d, w.hw_w = _param(dis.highway.weight, "highway.weight"); keep.append(d)
    d, w.hw_b = _param(dis.highway.bias, "highway.bias"); keep.append(d)
    d, w.out_w = _param(dis.feature2out.weight, "feature2out.weight"); keep.append(d)
    n_out = d.shape[0]
    d, w.out_b = _param(dis.feature2out.bias, "feature2out.bias"); keep.append(d)
    if keep[-4].shape != (F, F) or keep[-2].shape != (n_out, F):
        raise RuntimeError("textgan_hip: discriminator parameter shapes are inconsistent")
    w.V, w.Ed, w.F, w.n_out = V, DIS_EMBED_DIM, F, n_out
    return w, keep


def profile_enable(on):
    lib().tg_profile_enable(1 if on else 0)


def profile_read():
    """{kernel name: (total_ms, launches)} since the last read; synchronises on the recorded events."""
    out = {}
    for name, kid in KERNELS.items():
        ms, n = C.c_double(0.0), C.c_int(0)
        check(lib().tg_profile_read(kid, C.byref(ms), C.byref(n)))
        out[name] = (ms.value, n.value)
    return out


_seed_state = {"base": None, "count": 0}


def next_dropout_seed(rank=0):
    """A fresh 32-bit seed for the scorer's counter-based dropout masks (TgDisWeights.dropout_seed): derived from
    torch's CPU seed at first use (so torch.manual_seed makes runs repeatable), a call counter and the rank."""
    if _seed_state["base"] is None:
        _seed_state["base"] = int(torch.initial_seed()) & 0xffffffffffffffff
    _seed_state["count"] += 1
    x = (_seed_state["base"] + 0x9E3779B97F4A7C15 * (_seed_state["count"] + (rank << 40))) & 0xffffffffffffffff
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xffffffffffffffff   # splitmix64 finaliser
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xffffffffffffffff
    return int((x ^ (x >> 31)) & 0xffffffff)
