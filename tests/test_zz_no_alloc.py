"""tg_gen_sample_static -- the entry point that allocates nothing (its scratch is a static device array) -- against
tg_gen_sample with a caller-provided workspace: identical tokens, through the C ABI, on two calls in a row, the
second with a smaller input (fewer rows, shorter sequences, smaller vocabulary: what the first call left in the shared
array must not leak into it).  Run last (file name): a fault here must not hide the other GPU tests."""
import ctypes as C

import numpy as np
import pytest
import torch

import textgan_pytorch_amd as tg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(V, N, L, seed, forced=False):
    lib = tg._lib.lib()
    gen = tg.modules.synthetic_generator(V, 32, 32, L, seed=seed, std=1.0, device=DEV)
    w, keep = tg._lib.gen_weights(gen, "split_f16")
    u = torch.from_numpy(tg.synth.uniform01(seed + 40, (L, N))).to(DEV)
    f = torch.from_numpy(tg.synth.tokens(seed + 41, (N, L), V)).to(DEV) if forced else None
    fp = None if f is None else f.data_ptr()
    stream = tg._lib.stream_ptr()
    # existing entry point: caller-provided workspace
    nbytes = lib.tg_gen_sample_workspace_bytes(C.byref(w), N)
    assert 0 < nbytes <= lib.tg_gen_sample_static_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    want = torch.full((N, L), -1, dtype=torch.int64, device=DEV)
    tg._lib.check(lib.tg_gen_sample(C.byref(w), N, L, 1, u.data_ptr(), fp, want.data_ptr(), ws.data_ptr(), nbytes, stream))
    # the no-allocation entry point: no workspace argument at all.  One warm-up call first (code-object loading, the
    # symbol-address lookup and the LDS attribute happen on the first call on a device), then the measured call:
    # neither torch's caching allocator (memory_allocated) nor the DEVICE's free memory (mem_get_info: a hipMalloc /
    # hipFree inside libtextgan_hip.so never shows in torch's counter) may move across it.
    warm = torch.full((N, L), -3, dtype=torch.int64, device=DEV)
    tg._lib.check(lib.tg_gen_sample_static(C.byref(w), N, L, 1, u.data_ptr(), fp, warm.data_ptr(), stream))
    got = torch.full((N, L), -2, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    tg._lib.check(lib.tg_gen_sample_static(C.byref(w), N, L, 1, u.data_ptr(), fp, got.data_ptr(), stream))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before, "device free memory moved across tg_gen_sample_static"
    assert torch.equal(warm, got)
    return got.cpu().numpy(), want.cpu().numpy()


def test_static_scratch_sample_matches_workspace_sample():
    lib = tg._lib.lib()
    assert lib.tg_gen_sample_static_bytes() == 256 + 4 * 128 * 5 * 64 * 16
    got, want = _both(5255, 300, 20, seed=3)                  # first call: the real-data vocabulary
    assert got.min() >= 0 and got.max() < 5255
    np.testing.assert_array_equal(got, want)
    got2, want2 = _both(1000, 70, 12, seed=4)                 # second call in a row, smaller in every dimension
    assert got2.max() < 1000
    np.testing.assert_array_equal(got2, want2)
    got3, want3 = _both(16384, 33, 6, seed=5, forced=True)    # the largest vocabulary the array holds, replay mode
    np.testing.assert_array_equal(got3, want3)


def test_static_scratch_rejects_what_it_cannot_hold():
    lib = tg._lib.lib()
    out = torch.empty(4, 5, dtype=torch.int64, device=DEV)
    u = torch.rand(5, 4, device=DEV)
    big = tg.modules.synthetic_generator(16385, 32, 32, 5, device=DEV)
    w, keep = tg._lib.gen_weights(big, "split_f16")
    assert lib.tg_gen_sample_static(C.byref(w), 4, 5, 1, u.data_ptr(), None, out.data_ptr(), tg._lib.stream_ptr()) == -1
    assert b"exceeds the static scratch" in lib.tg_last_error()
    small = tg.modules.synthetic_generator(100, 32, 32, 5, device=DEV)
    w, keep = tg._lib.gen_weights(small, "fp32")              # needs no scratch: pointed at tg_gen_sample
    assert lib.tg_gen_sample_static(C.byref(w), 4, 5, 1, u.data_ptr(), None, out.data_ptr(), tg._lib.stream_ptr()) == -1
    wide = tg.modules.synthetic_generator(100, 64, 64, 5, device=DEV)
    w, keep = tg._lib.gen_weights(wide, "split_f16")
    assert lib.tg_gen_sample_static(C.byref(w), 4, 5, 1, u.data_ptr(), None, out.data_ptr(), tg._lib.stream_ptr()) == -1
    assert b"E = H = 32" in lib.tg_last_error()
