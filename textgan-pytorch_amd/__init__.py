"""textgan-pytorch_amd: MI355X-native Monte-Carlo rollout reward path
(drop-in for TextGAN-PyTorch's utils/rollout.py ROLLOUT.get_reward, plus
LSTMGenerator.sample and NLL.cal_nll), hand-written HIP behind a C ABI."""
from . import synth  # noqa: F401

__all__ = ["synth"]
