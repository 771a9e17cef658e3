"""Policy-gradient loss of the adversarial generator step on the HIP kernels.

Drop-in for ``SeqGAN_G.batchPGLoss`` (models/SeqGAN_G.py:21-39), ``MaliGAN_G.adv_loss``
(models/MaliGAN_G.py:21-39) and ``SentiGAN_G.batchPGLoss`` (models/SentiGAN_G.py:47-65): the returned scalar
carries an autograd node whose backward fills ``.grad`` of the generator's seven parameters, so the instructor's
``self.optimize(opt, adv_loss)`` (instructor/real_data/seqgan_instructor.py:109-110) works unchanged.  Neither the
[B*L, V] log-softmax matrix nor the [B, L, V] one-hot of the reference is ever materialised (E = H = 32: fused
kernels, csrc/tg_pg.hip; any other E, H multiple of 32 up to 1024: GEMMs over <= 1 GiB of logits at a time,
csrc/tg_pgw.hip -- the three vocabulary products as split-fp16 x3 GEMMs by default, ``precision="fp32"`` for exact
fp32 MFMA).
"""
import torch

from . import ops

_PARAMS = ("embeddings.weight", "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
           "lstm2out.weight", "lstm2out.bias")


def _params(gen):
    return (gen.embeddings.weight, gen.lstm.weight_ih_l0, gen.lstm.weight_hh_l0, gen.lstm.bias_ih_l0,
            gen.lstm.bias_hh_l0, gen.lstm2out.weight, gen.lstm2out.bias)


class _PGLoss(torch.autograd.Function):
    """loss = -sum(pred * weight) with pred = log p(target) (use_prob False) or p(target) (True)."""

    @staticmethod
    def forward(ctx, gen, inp, target, weight, use_prob, padding_idx, precision, *params):
        r = ops.gen_pg_forward(gen, inp, target, precision=precision)
        logp = r["logp"]
        pred = torch.exp(logp) if use_prob else logp
        ctx.gen, ctx.use_prob, ctx.padding_idx, ctx.precision = gen, use_prob, padding_idx, precision
        ctx.save_for_backward(inp, target, weight, logp, r["lse"], r["states"])
        return -(pred * weight).sum()

    @staticmethod
    def backward(ctx, grad_out):
        inp, target, weight, logp, lse, states = ctx.saved_tensors
        coeff = -weight * grad_out
        if ctx.use_prob:
            coeff = coeff * torch.exp(logp)
        grads = ops.gen_pg_backward(ctx.gen, inp, target, coeff.float().contiguous(), lse, states, ctx.padding_idx,
                                    precision=ctx.precision)
        return (None,) * 7 + tuple(grads)


def pg_loss(gen, inp, target, weight, use_prob=False, precision=None):
    """-sum(pred[n,t] * weight[n,t]); ``weight`` broadcasts against [N,L] like the reference's ``pred * reward``.
    precision: None = the library default ("split_f16"), or "fp32"; only shapes beyond E = H = 32 have the choice."""
    dev = gen.lstm2out.weight.device
    inp, target = inp.to(dev).long().contiguous(), target.to(dev).long().contiguous()
    if inp.dim() != 2 or inp.shape != target.shape:
        raise RuntimeError("batchPGLoss: inp and target must both be [batch, seq_len], got %s and %s"
                           % (tuple(inp.shape), tuple(target.shape)))
    weight = torch.broadcast_to(weight.to(dev).float(), inp.shape).contiguous()
    pad = getattr(gen, "padding_idx", None)
    if pad is None:
        pad = gen.embeddings.padding_idx
    return _PGLoss.apply(gen, inp, target, weight.detach(), bool(use_prob), -1 if pad is None else int(pad),
                         precision, *_params(gen))


def batchPGLoss(gen, inp, target, reward, precision=None):
    """SeqGAN_G.batchPGLoss (models/SeqGAN_G.py:21-39): -sum(logp(target) * reward); for a generator whose
    ``name`` is 'sentigan', SentiGAN_G.batchPGLoss (models/SentiGAN_G.py:47-65): -sum(p(target) * (1 - reward))."""
    if getattr(gen, "name", "") == "sentigan":
        return pg_loss(gen, inp, target, 1.0 - reward, use_prob=True, precision=precision)
    return pg_loss(gen, inp, target, reward, use_prob=False, precision=precision)


def adv_loss(gen, inp, target, reward, precision=None):
    """MaliGAN_G.adv_loss (models/MaliGAN_G.py:21-39): the same expression as SeqGAN's."""
    return pg_loss(gen, inp, target, reward, use_prob=False, precision=precision)
