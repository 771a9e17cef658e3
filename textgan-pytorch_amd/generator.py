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
