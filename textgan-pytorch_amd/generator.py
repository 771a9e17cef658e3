"""Same-kernel surfaces of ``LSTMGenerator`` (models/generator.py:40-84)."""
import torch

from . import ops


def sample(gen, num_samples, batch_size, start_letter=1, uniforms=None, generator=None, device=None):
    """LSTMGenerator.sample (models/generator.py:62-84): ``num_samples`` sequences of length
    ``gen.max_seq_len``, returned as a CPU int64 tensor like the reference (its buffer is never
    moved to the GPU, :68,:82).  The reference samples ``num_batch`` batches and truncates
    (:67,:82); here all num_batch*batch_size rows run in one launch.

    uniforms: optional [L, num_batch*batch_size] pre-drawn U[0,1)."""
    L = gen.max_seq_len
    num_batch = num_samples // batch_size + 1 if num_samples != batch_size else 1  # generator.py:67
    N = num_batch * batch_size
    dev = device if device is not None else gen.lstm2out.weight.device
    with torch.no_grad():
        u = uniforms if uniforms is not None else torch.rand(L, N, dtype=torch.float32, device=dev,
                                                             generator=generator)
        out = ops.gen_sample(gen, N, L, start_letter, u)
    return out[:num_samples].cpu()


def forward_logprobs(gen, inp):
    """LSTMGenerator.forward(inp, init_hidden(N)) for a [N,L] input (models/generator.py:40-60):
    log-probabilities [(N*L), V], row = n*L + t.  Parity surface; materialises the logits."""
    with torch.no_grad():
        r = ops.gen_teacher_forced(gen, inp, want_lse=True, want_logits=True)
    N, L, V = r["logits"].shape
    return (r["logits"] - r["lse"].unsqueeze(-1)).view(N * L, V)


def forward(gen, inp, hidden=None, need_hidden=False):
    """LSTMGenerator.forward(inp, hidden, need_hidden) (models/generator.py:40-60) under no_grad: ``inp`` [N] (one
    step, :48-49) or [N,T]; ``hidden`` None (zeros, init_hidden) or any carried (h, c), each [1,N,H], as the reference
    accepts; returns log-probabilities [(N*T), V] (row = n*T + t) and, with ``need_hidden``, the new (h, c)."""
    if inp.dim() == 1:
        inp = inp.unsqueeze(1)
    states_in = None
    if hidden is not None:
        h, c = hidden
        N = inp.shape[0]
        if h.dim() != 3 or c.dim() != 3 or h.shape[0] != 1 or h.shape[1] != N or tuple(h.shape) != tuple(c.shape):
            raise RuntimeError("textgan_hip generator.forward: hidden must be (h, c), each [1, %d, H], got %s and %s"
                               % (N, tuple(h.shape), tuple(c.shape)))
        states_in = torch.stack([h[0], c[0]], 1).to(device=inp.device, dtype=torch.float32).contiguous()   # [N,2,H]
    with torch.no_grad():
        r = ops.gen_teacher_forced(gen, inp, want_lse=True, want_logits=True, want_states=need_hidden,
                                   states_in=states_in)
    N, T, V = r["logits"].shape
    pred = (r["logits"] - r["lse"].unsqueeze(-1)).view(N * T, V)
    if not need_hidden:
        return pred
    last = r["states"][T]                       # [N, 2, H]: (h, c) after the last position
    return pred, (last[:, 0].unsqueeze(0).contiguous(), last[:, 1].unsqueeze(0).contiguous())
