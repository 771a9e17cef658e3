"""textgan-pytorch_amd: MI355X-native Monte-Carlo rollout reward path
(drop-in for TextGAN-PyTorch's utils/rollout.py ROLLOUT.get_reward, plus
LSTMGenerator.sample and NLL.cal_nll), hand-written HIP behind a C ABI.

    from textgan_pytorch_amd import ROLLOUT          # same API as utils.rollout.ROLLOUT
    rewards = ROLLOUT(gen).get_reward(target, rollout_num, dis)
"""
from . import synth  # noqa: F401  (numpy only)

__all__ = ["synth", "ROLLOUT", "install", "sample", "cal_nll", "modules", "ops", "dist"]


def __getattr__(name):  # torch-dependent parts load lazily
    import importlib
    if name in ("modules", "ops", "dist", "rollout", "generator", "nll", "discriminator", "maligan", "pgloss", "_lib"):
        return importlib.import_module("." + name, __name__)
    if name in ("ROLLOUT", "install"):
        return getattr(importlib.import_module(".rollout", __name__), name)
    if name == "sample":
        return importlib.import_module(".generator", __name__).sample
    if name == "cal_nll":
        return importlib.import_module(".nll", __name__).cal_nll
    raise AttributeError(name)
