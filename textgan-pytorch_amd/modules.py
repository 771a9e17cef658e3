"""State_dict-compatible shells of the reference's generator / discriminator
modules (parameter names and shapes only), for benchmarks and tests on machines
where the reference tree is absent.  A real training run keeps using the
reference's own classes -- the HIP path reads any module with these attributes.

  LSTMGenerator     models/generator.py:18-38   (SeqGAN_G / MaliGAN_G / SentiGAN_G / Oracle: :16-19 of each)
  LeakGAN_G         models/LeakGAN_G.py:23-50
  CNNDiscriminator  models/discriminator.py:20-38
  SeqGAN_D          models/SeqGAN_D.py:12-19    (MaliGAN_D identical)
  SentiGAN_D        models/SentiGAN_D.py:14-28
"""
import numpy as np
import torch
import torch.nn as nn

from . import synth


class LSTMGenerator(nn.Module):
    def __init__(self, embedding_dim, hidden_dim, vocab_size, max_seq_len, padding_idx=0, gpu=True, name="vanilla"):
        super().__init__()
        self.name = name
        self.embedding_dim, self.hidden_dim = embedding_dim, hidden_dim
        self.vocab_size, self.max_seq_len = vocab_size, max_seq_len
        self.padding_idx, self.gpu = padding_idx, gpu
        self.embeddings = nn.Embedding(vocab_size, embedding_dim, padding_idx=padding_idx)
        self.lstm = nn.LSTM(embedding_dim, hidden_dim, batch_first=True)
        self.lstm2out = nn.Linear(hidden_dim, vocab_size)

    def init_hidden(self, batch_size):
        p = self.lstm2out.weight
        z = torch.zeros(1, batch_size, self.hidden_dim, device=p.device, dtype=p.dtype)
        return z, z.clone()


class LeakGAN_G(nn.Module):
    """State_dict-compatible shell of models/LeakGAN_G.py:23-50 (parameters and the attributes the rollout reads)."""

    def __init__(self, embedding_dim, hidden_dim, vocab_size, max_seq_len, padding_idx=0, goal_size=16, step_size=4,
                 gpu=True, goal_out_size=sum(synth.SEQGAN_D_FILTERS[1]), batch_size=64):
        super().__init__()
        self.name = "leakgan"
        self.hidden_dim, self.embedding_dim = hidden_dim, embedding_dim
        self.max_seq_len, self.vocab_size, self.padding_idx = max_seq_len, vocab_size, padding_idx
        self.goal_size, self.goal_out_size, self.step_size = goal_size, goal_out_size, step_size
        self.gpu, self.temperature = gpu, 1.5
        self.embeddings = nn.Embedding(vocab_size, embedding_dim, padding_idx=padding_idx)
        self.worker = nn.LSTM(embedding_dim, hidden_dim)
        self.manager = nn.LSTM(goal_out_size, hidden_dim)
        self.work2goal = nn.Linear(hidden_dim, vocab_size * goal_size)
        self.mana2goal = nn.Linear(hidden_dim, goal_out_size)
        self.goal2goal = nn.Linear(goal_out_size, goal_size, bias=False)
        self.goal_init = nn.Parameter(torch.rand((batch_size, goal_out_size)))


class CNNDiscriminator(nn.Module):
    def __init__(self, embed_dim, vocab_size, filter_sizes, num_filters, padding_idx=0, gpu=True, dropout=0.2,
                 n_out=2):
        super().__init__()
        self.embedding_dim, self.vocab_size, self.padding_idx = embed_dim, vocab_size, padding_idx
        self.feature_dim = int(sum(num_filters))
        self.gpu = gpu
        self.embeddings = nn.Embedding(vocab_size, embed_dim, padding_idx=padding_idx)
        self.convs = nn.ModuleList([nn.Conv2d(1, n, (f, embed_dim)) for n, f in zip(num_filters, filter_sizes)])
        self.highway = nn.Linear(self.feature_dim, self.feature_dim)
        self.feature2out = nn.Linear(self.feature_dim, n_out)
        self.dropout = nn.Dropout(dropout)


def SeqGAN_D(embed_dim, vocab_size, padding_idx=0, gpu=True, dropout=0.25):
    return CNNDiscriminator(embed_dim, vocab_size, *synth.SEQGAN_D_FILTERS, padding_idx, gpu, dropout, 2)


def SentiGAN_D(k_label, embed_dim, vocab_size, padding_idx=0, gpu=True, dropout=0.2):
    return CNNDiscriminator(embed_dim, vocab_size, *synth.SENTIGAN_D_FILTERS, padding_idx, gpu, dropout, k_label + 1)


def load_numpy_state(module, sd, device=None):
    """Load a numpy state_dict (synth.gen_state / synth.dis_state) into a module."""
    module.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return module.to(device) if device is not None else module


def synthetic_generator(V, E, H, L, seed=0, std=1.0, device="cuda"):
    g = LSTMGenerator(E, H, V, L)
    return load_numpy_state(g, synth.gen_state(V, E, H, seed=seed, std=std), device)


def synthetic_leakgan(V, E, H, L, goal_size=16, step_size=4, batch_size=64, seed=0, std=0.1, device="cuda"):
    g = LeakGAN_G(E, H, V, L, goal_size=goal_size, step_size=step_size, batch_size=batch_size)
    return load_numpy_state(g, synth.leak_state(V, E, H, goal_size, g.goal_out_size, batch_size, seed=seed, std=std), device)


def synthetic_discriminator(kind, V, seed=0, k_label=4, device="cuda"):
    if kind == "seqgan":
        d = SeqGAN_D(64, V)
        sd = synth.dis_state(V, 64, *synth.SEQGAN_D_FILTERS, n_out=2, seed=seed)
    elif kind == "sentigan":
        d = SentiGAN_D(k_label, 64, V)
        sd = synth.dis_state(V, 64, *synth.SENTIGAN_D_FILTERS, n_out=k_label + 1, seed=seed)
    else:
        raise ValueError(kind)
    return load_numpy_state(d, sd, device)
