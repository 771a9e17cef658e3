"""Same-kernel surfaces of ``CNNDiscriminator`` (models/discriminator.py:40-64)."""
import torch

from . import ops


def forward(dis, inp, dropout_uniforms=None):
    """CNNDiscriminator.forward: logits [N, n_out].  Dropout follows dis.training like nn.Dropout."""
    p = float(dis.dropout.p) if (dis.training and dis.dropout.p > 0) else 0.0
    with torch.no_grad():
        # without explicit uniforms the masks come from the kernel's counter-based generator (a fresh seed per call)
        return ops.dis_score(dis, inp, dropout_p=p, dropout_uniforms=dropout_uniforms, want_reward=False,
                             want_logits=True)["logits"]


def get_feature(dis, inp):
    """CNNDiscriminator.get_feature: highway features [N, F]."""
    with torch.no_grad():
        return ops.dis_score(dis, inp, want_reward=False, want_feature=True)["feature"]
