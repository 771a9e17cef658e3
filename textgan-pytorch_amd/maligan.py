"""MaliGAN reward (instructor/real_data/maligan_instructor.py:142-154, oracle_data twin): SURVEY 8(f)3.

The reference runs ``cfg.rollout_num`` full forward passes of D over the same samples; they differ only in their
dropout masks.  Here the conv + highway features are computed once and the R masks, the output layer and the softmax
column run in one small kernel (tg_dis_rescore_dropout)."""
import torch

from . import _lib, ops


def get_mali_reward(dis, samples, rollout_num, dropout_uniforms=None, generator=None, precision=None):
    """MaliGANInstructor.get_mali_reward(samples) with ``self.dis = dis``, ``cfg.rollout_num = rollout_num``.

    samples [B,L] int64 on the GPU; returns [B,L] fp32 (every column equal, like the reference's expand).
    dropout_uniforms: optional [R,B,F] U[0,1) (mask r keeps u < 1-p); without them the masks are drawn with
    torch.rand if a ``generator`` is given, else by the kernel's counter-based generator.
    Dropout follows ``dis.training`` as nn.Dropout does (the reference never calls dis.eval())."""
    if not samples.is_cuda:
        raise RuntimeError("textgan_hip get_mali_reward: samples must be on the GPU (got %s); there is no CPU path"
                           % samples.device)
    R = int(rollout_num)
    B = samples.shape[0]
    with torch.no_grad():
        weights = _lib.dis_weights(dis, precision)
        feat = ops.dis_score(dis, samples, want_reward=False, want_feature=True, weights=weights)["feature"]
        p = float(dis.dropout.p) if (dis.training and dis.dropout.p > 0) else 0.0
        du = dropout_uniforms
        if p > 0 and du is None and generator is not None:
            du = torch.rand(R, B, weights[0].F, dtype=torch.float32, device=samples.device, generator=generator)
        # neither uniforms nor a torch.Generator: counter-based masks inside the kernel (a fresh seed per call)
        probs = ops.dis_rescore_dropout(dis, feat, R, col=1, dropout_p=p, dropout_uniforms=du, weights=weights)
        rewards = torch.mean(probs, dim=0)                 # maligan_instructor.py:149
        rewards = torch.div(rewards, 1 - rewards)          # :150
        rewards = torch.div(rewards, torch.sum(rewards))   # :151
        rewards = rewards - torch.mean(rewards)            # :152
        return rewards.unsqueeze(1).expand(samples.size())  # :153
