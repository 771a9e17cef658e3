"""CPU-side checks: the C-ABI library loads and exports every declared symbol, host logic
(shard plan, sample batching rule, synthetic generator determinism), error behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import textgan_pytorch_amd as tg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "textgan_hip.h")).read()
    declared = set(re.findall(r"\b(tg_[a-z_0-9]+)\s*\(", hdr))
    assert declared >= set(tg._lib.SYMBOLS)
    lib = ctypes.CDLL(tg._lib.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    lib.tg_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.tg_version()


def test_struct_layouts_match_header():
    # 7 pointers + 4 int32 / see TgDisWeights in include/textgan_hip.h
    assert ctypes.sizeof(tg._lib.GenW) == 7 * 8 + 16
    assert ctypes.sizeof(tg._lib.DisW) == 8 + 2 * 16 * 8 + 4 * 8 + 2 * 16 * 4 + 7 * 4 + 4  # 7 ints + tail padding


def test_no_cpu_fallback():
    gen = tg.modules.synthetic_generator(50, 32, 32, 20, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        tg.ROLLOUT(gen).get_reward(torch.zeros(2, 20, dtype=torch.long), 2, None)
    with pytest.raises(RuntimeError):
        tg.ops.gen_sample(gen, 4, 20, 1, torch.rand(20, 4))


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under the product package (Python or HIP sources, the Makefile)
    imports, loads, links or names it, nor the reference checkout."""
    pkg = os.path.dirname(os.path.abspath(tg._lib.__file__))
    bad = re.compile(r"\boracle\s*(import|\.|/)|import\s+oracle|from\s+oracle|libtextgan_oracle|/root/reference")
    hits = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                for i, line in enumerate(open(os.path.join(d, f), errors="replace"), 1):
                    # "oracle" also names the reference's Oracle generator (models/Oracle.py) and its data set
                    code = line.split("#")[0] if f.endswith(".py") else line.split("//")[0]   # comments may cite it
                    if bad.search(code):
                        hits.append("%s:%d: %s" % (os.path.join(d, f), i, line.strip()))
    assert not hits, "\n".join(hits)


def test_shells_and_rollout_match_the_reference_contract():
    """tests/golden/ref_contract.json (tools/make_ref_contract.py, written from the reference's own classes): the
    state_dict keys and shapes, the attributes the shim reads and ROLLOUT's public surface.  The shells the GPU tests
    and the bench build (textgan-pytorch_amd/modules.py) and tg.ROLLOUT are held to it."""
    import json
    c = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_contract.json")))
    sh = c["shape"]
    V, L, E, H, B = sh["V"], sh["L"], sh["E"], sh["H"], sh["batch_size"]

    def sd(m):
        return {k: list(v.shape) for k, v in m.state_dict().items()}

    for name in ("SeqGAN_G", "SentiGAN_G", "MaliGAN_G"):
        g = tg.modules.LSTMGenerator(E, H, V, L, 0, name=c[name]["attrs"]["name"])
        assert sd(g) == c[name]["state_dict"], name
        for k, v in c[name]["attrs"].items():
            assert getattr(g, k) == v, (name, k)
    lg = tg.modules.LeakGAN_G(E, H, V, L, 0, goal_size=16, step_size=4, batch_size=B)
    assert sd(lg) == c["LeakGAN_G"]["state_dict"]
    for k, v in c["LeakGAN_G"]["attrs"].items():
        assert getattr(lg, k) == v, k
    for name, d in (("SeqGAN_D", tg.modules.SeqGAN_D(64, V)), ("SentiGAN_D", tg.modules.SentiGAN_D(4, 64, V)),
                    ("MaliGAN_D", tg.modules.SeqGAN_D(64, V)),
                    ("LeakGAN_D", tg.modules.SeqGAN_D(64, V, dropout=c["LeakGAN_D"]["attrs"]["dropout_p"]))):
        a = c[name]["attrs"]
        assert sd(d) == c[name]["state_dict"], name
        assert (d.feature_dim == a["feature_dim"] and d.dropout.p == a["dropout_p"] and d.training == a["training_after_init"]
                and [cv.kernel_size[0] for cv in d.convs] == a["filter_widths"]
                and [cv.out_channels for cv in d.convs] == a["filter_counts"]
                and d.feature2out.out_features == a["n_out"] and d.embedding_dim == a["embed_dim"]), name
    ro = tg.ROLLOUT(lg, gpu=False)
    for k in c["ROLLOUT"]["attrs"]:
        assert hasattr(ro, k), k
    assert (ro.step_size, ro.goal_out_size) == (4, c["LeakGAN_G"]["attrs"]["goal_out_size"])
    for k in c["ROLLOUT"]["methods"]:
        assert callable(getattr(ro, k)), k


def test_install_binds_the_optional_surfaces():
    """tg.install on stand-ins shaped like the reference's modules and classes: ROLLOUT replaced in the module, sample /
    cal_nll / batchPGLoss / adv_loss bound on the classes with the reference's signatures, originals kept."""
    import types
    mod = types.SimpleNamespace(ROLLOUT=object)

    class Gen(torch.nn.Module):
        def sample(self, num_samples, batch_size, start_letter=1):
            return "ref sample"

    class NLL:
        @staticmethod
        def cal_nll(model, data_loader, criterion, gpu=True):
            return "ref nll"

    class SeqG(Gen):
        def batchPGLoss(self, inp, target, reward):
            return "ref pg"

    class MaliG(Gen):
        def adv_loss(self, inp, target, reward):
            return "ref adv"

    assert tg.install(mod, Gen, NLL, (SeqG, MaliG)) is tg.ROLLOUT and mod.ROLLOUT is tg.ROLLOUT
    assert Gen._tg_original_sample(Gen(), 1, 1) == "ref sample" and NLL._tg_original_cal_nll(None, None, None) == "ref nll"
    assert SeqG._tg_original_batchPGLoss(SeqG(), 0, 0, 0) == "ref pg" and MaliG._tg_original_adv_loss(MaliG(), 0, 0, 0) == "ref adv"
    assert "adv_loss" not in SeqG.__dict__ and "batchPGLoss" not in MaliG.__dict__
    g = tg.modules.synthetic_generator(30, 32, 32, 20, device="cpu")
    g.__class__ = type("G2", (Gen,), {})            # a CPU generator: the bound methods reach the HIP path and refuse
    with pytest.raises(RuntimeError):
        g.sample(4, 4)
    with pytest.raises(RuntimeError):
        NLL.cal_nll(g, [{"input": torch.zeros(2, 20, dtype=torch.long), "target": torch.zeros(2, 20, dtype=torch.long)}], None)


def test_rollout_attributes_mirror_reference():
    gen = tg.modules.synthetic_generator(50, 32, 32, 20, device="cpu")
    ro = tg.ROLLOUT(gen, gpu=True)
    assert (ro.gen is gen and ro.max_seq_len == 20 and ro.vocab_size == 50 and ro.step_size == 0
            and ro.goal_out_size == 0 and ro.gpu is True)
    assert ro.old_model is not gen and ro.old_model.vocab_size == 50  # lazy deep copy (rollout.py:18)


def test_shard_plan():
    P = tg.dist.ShardPlan
    for R in (1, 7, 16, 64):
        for W in (1, 2, 3, 4, 8):
            plans = [P(R, W, r) for r in range(W)]
            assert plans[0].i_lo == 0 and plans[-1].i_hi == R
            assert all(a.i_hi == b.i_lo for a, b in zip(plans, plans[1:]))
            assert max(p.i_hi - p.i_lo for p in plans) - min(p.i_hi - p.i_lo for p in plans) <= 1
            assert plans[0].even == (R % W == 0)
    with pytest.raises(ValueError):
        P(4, 2, 2)


def test_synth_is_deterministic_and_shaped():
    a, b = tg.synth.gen_state(100, 32, 32, seed=3), tg.synth.gen_state(100, 32, 32, seed=3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["lstm.weight_ih_l0"].shape == (128, 32) and a["lstm2out.weight"].shape == (100, 32)
    d = tg.synth.dis_state(100, 64, *tg.synth.SEQGAN_D_FILTERS)
    assert d["convs.11.weight"].shape == (160, 1, 20, 64) and d["highway.weight"].shape == (1720, 1720)
    assert float(np.abs(d["highway.weight"]).max()) <= 0.0500001
    u = tg.synth.uniform01(1, (1000,))
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1
    g = tg.modules.synthetic_generator(100, 32, 32, 20, device="cpu")
    assert sorted(g.state_dict()) == sorted(a)


def test_wide_generator_workspace_is_bounded_by_the_chunk():
    """tg_gen_rollout_workspace_bytes (host arithmetic only): E = H = 32 needs the prefix states and the split W_out
    fragments; other shapes in split_f16 add one chunk of at most 2^18 completions of the GEMM path (tg_genw.hip),
    whatever rollout_num is."""
    import ctypes as C
    lib = tg._lib.lib()

    def ws(V, E, H, B, L, R, prec):
        w = tg._lib.GenW(*([1] * 7), V, E, H, tg._lib.PRECISIONS[prec])  # pointers are not dereferenced
        return lib.tg_gen_rollout_workspace_bytes(C.byref(w), B, L, R)

    small = ws(5255, 32, 32, 64, 51, 16, "split_f16")
    assert small < 4 << 20 and small == ws(5255, 32, 32, 64, 51, 64, "split_f16")
    one, many = ws(32000, 512, 512, 4096, 51, 1, "split_f16"), ws(32000, 512, 512, 4096, 51, 64, "split_f16")
    assert one == many and 4 << 30 < one < 8 << 30          # one rollout index = 208 896 rows is already a chunk
    assert ws(32000, 512, 512, 4096, 51, 64, "fp32") < 1 << 30   # persistent kernel: prefix states only
    assert ws(3000, 64, 128, 8, 20, 2, "split_f16") < ws(3000, 64, 128, 8, 20, 200, "split_f16") \
        <= ws(3000, 64, 128, 8, 20, 100000, "split_f16") < 2 << 30


def test_allgather_rows_rejects_what_it_cannot_do():
    """tg_allgather_rows needs a caller-owned RCCL communicator; without one (or without RCCL in the process) it
    reports an error instead of touching memory."""
    lib = tg._lib.lib()
    assert lib.tg_allgather_rows(None, 4, 8, None, None) == -1
    assert b"bad arguments" in lib.tg_last_error()
    buf = (ctypes.c_float * 64)()
    assert lib.tg_allgather_rows(buf, 0, 8, ctypes.c_void_p(1), None) == -1


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the form of the driver's N = 1 command) starts
    its ranks itself through torch.distributed.run, as a child process, and relays rank 0's one JSON line; under a
    launcher (WORLD_SIZE set) it is a rank.  TG_BENCH_DRYRUN=1: rendezvous and relay only, no GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TG_BENCH_DRYRUN"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # stdout carries the one JSON line and nothing else
    rec = json.loads(lines[0])
    assert rec["dry_run"] is True and rec["n_gpus"] == 2
    assert b"torch.distributed.run" in p.stderr         # it said what it started
    # a mismatch between --gpus and the launcher's world size is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p2.returncode != 0 and b"WORLD_SIZE=1" in p2.stderr
