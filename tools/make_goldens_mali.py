#!/usr/bin/env python3
"""Golden vectors for the MaliGAN reward (instructor/real_data/maligan_instructor.py:142-154).

The instructor module itself does not import here (instructor.py -> metrics/bleu.py needs nltk.translate), so the
method `MaliGANInstructor.get_mali_reward` is taken from the reference file's syntax tree and executed as it
stands, against the reference's own MaliGAN_D, with `dis.dropout` replaced on the instance by the mask-from-uniform
module of tools/make_goldens.py (call r uses uniforms[r]).  Nothing of the reference is written to the repo: the
.npz holds seeds, shapes and outputs.  Run from a directory other than /root/reference:
    python tools/make_goldens_mali.py
"""
import ast
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TEXTGAN_REFERENCE", "/root/reference")
sys.modules.setdefault("nltk", types.ModuleType("nltk"))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import config as cfg  # noqa: E402
cfg.CUDA = False
cfg.dis_init = "uniform"
from models.MaliGAN_D import MaliGAN_D  # noqa: E402

import textgan_pytorch_amd as tg  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_goldens import MaskDropout, to_torch_sd  # noqa: E402

synth = tg.synth
OUT = os.path.join(ROOT, "tests", "golden")


def reference_method(path, cls, name, namespace):
    """Compile one method of a class straight from the reference source file."""
    tree = ast.parse(open(path).read(), path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == name:
                    mod = ast.Module(body=[fn], type_ignores=[])
                    exec(compile(mod, path, "exec"), namespace)
                    return namespace[name]
    raise KeyError(name)


def run(name, V, L, B, R, dseed, sseed, useed):
    dis = MaliGAN_D(64, V, 0, gpu=False)
    dis.load_state_dict(to_torch_sd(synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=dseed)))
    p = float(dis.dropout.p)
    du = synth.uniform01(useed, (R, B, dis.feature_dim))
    dis.dropout = MaskDropout(p, du)
    samples = synth.tokens(sseed, (B, L), V)
    cfg.rollout_num = R
    fn = reference_method(os.path.join(REF, "instructor", "real_data", "maligan_instructor.py"),
                          "MaliGANInstructor", "get_mali_reward", {"torch": torch, "F": F, "cfg": cfg})
    with torch.no_grad():
        rew = fn(types.SimpleNamespace(dis=dis), torch.from_numpy(samples))
    assert dis.dropout.c == R and tuple(rew.shape) == (B, L)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="mali_reward", V=V, L=L, B=B, R=R, dseed=dseed,
                        sseed=sseed, useed=useed, dropout_p=p, reward=rew.numpy().astype(np.float32))
    print(name, "reward[:,0] range", float(rew.min()), float(rew.max()))


if __name__ == "__main__":
    run("mali_tiny", 200, 20, 8, 4, 111, 112, 113)
    run("mali_b64_r16", 5000, 20, 64, 16, 114, 115, 116)
