"""Deterministic synthetic inputs (weights, token ids, uniforms).

A build-owned integer-hash generator: every value is a pure function of
(seed, flat index) computed with exact uint64 / float64 arithmetic, so the
build container and the GPU box regenerate bit-identical tensors without any
file travelling between them.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def hash_u64(seed, n):
    """splitmix64 of (seed, index) for index in [0, n) -> uint64[n]."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03) + _GOLD
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def uniform01(seed, shape):
    """fp32 uniforms in [0,1) with 24-bit resolution (like torch.rand)."""
    n = int(np.prod(shape))
    h = hash_u64(seed, n) >> np.uint64(40)
    return (h.astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32).reshape(shape)


def uniform_sym(seed, shape, a):
    """fp32 uniform in [-a, a)."""
    return ((uniform01(seed, shape).astype(np.float64) * 2.0 - 1.0) * a).astype(np.float32)


def normal_like(seed, shape, std=1.0):
    """Zero-mean, unit-variance (x std) Irwin-Hall(4) variate, exact in float64."""
    n = int(np.prod(shape))
    h = hash_u64(seed, n)
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + ((h >> np.uint64(48)) & m))
    z = (s.astype(np.float64) / 65536.0 - 2.0) * np.sqrt(3.0) * std
    return z.astype(np.float32).reshape(shape)


def tokens(seed, shape, vocab):
    """int64 token ids uniform in [0, vocab)."""
    n = int(np.prod(shape))
    return (hash_u64(seed, n) % np.uint64(vocab)).astype(np.int64).reshape(shape)


def gen_state(V, E, H, seed=0, std=1.0):
    """LSTMGenerator state_dict as numpy arrays (models/generator.py:33-35 keys).

    std=1.0 mimics init_oracle (generator.py:97-100, N(0,1) everywhere)."""
    shapes = [("embeddings.weight", (V, E)), ("lstm.weight_ih_l0", (4 * H, E)),
              ("lstm.weight_hh_l0", (4 * H, H)), ("lstm.bias_ih_l0", (4 * H,)),
              ("lstm.bias_hh_l0", (4 * H,)), ("lstm2out.weight", (V, H)), ("lstm2out.bias", (V,))]
    return {k: normal_like(seed * 1000 + 17 * i + 1, s, std) for i, (k, s) in enumerate(shapes)}


def leak_state(V, E, H, K=16, G=1720, B_init=64, seed=0, std=0.1):
    """LeakGAN_G state_dict as numpy arrays (models/LeakGAN_G.py:40-48 keys): worker / manager LSTMs, work2goal
    [V*K, H], mana2goal [G, H], goal2goal [K, G] (no bias), goal_init [B_init, G] ~ U[0,1) (torch.rand, :48).
    G = goal_out_size = the discriminator's feature width, K = goal_size."""
    shapes = [("embeddings.weight", (V, E)),
              ("worker.weight_ih_l0", (4 * H, E)), ("worker.weight_hh_l0", (4 * H, H)),
              ("worker.bias_ih_l0", (4 * H,)), ("worker.bias_hh_l0", (4 * H,)),
              ("manager.weight_ih_l0", (4 * H, G)), ("manager.weight_hh_l0", (4 * H, H)),
              ("manager.bias_ih_l0", (4 * H,)), ("manager.bias_hh_l0", (4 * H,)),
              ("work2goal.weight", (V * K, H)), ("work2goal.bias", (V * K,)),
              ("mana2goal.weight", (G, H)), ("mana2goal.bias", (G,)),
              ("goal2goal.weight", (K, G))]
    sd = {k: normal_like(seed * 1000 + 701 + 13 * i, s, std) for i, (k, s) in enumerate(shapes)}
    sd["goal_init"] = uniform01(seed * 1000 + 799, (B_init, G))
    return sd


SEQGAN_D_FILTERS = ([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 20],
                    [100, 200, 200, 200, 200, 100, 100, 100, 100, 100, 160, 160])  # models/SeqGAN_D.py:12-13
SENTIGAN_D_FILTERS = ([2, 3, 4, 5], [200, 200, 200, 200])  # models/SentiGAN_D.py:14-15


def dis_state(V, Ed, filter_sizes, num_filters, n_out=2, seed=0, a=0.05):
    """CNNDiscriminator state_dict as numpy arrays, U(-a,a) like cfg.dis_init='uniform'
    (models/discriminator.py:70-71)."""
    F = int(sum(num_filters))
    sd = {"embeddings.weight": uniform_sym(seed * 1000 + 501, (V, Ed), a)}
    for k, (f, n) in enumerate(zip(filter_sizes, num_filters)):
        sd["convs.%d.weight" % k] = uniform_sym(seed * 1000 + 510 + 2 * k, (n, 1, f, Ed), a)
        sd["convs.%d.bias" % k] = uniform_sym(seed * 1000 + 511 + 2 * k, (n,), a)
    sd["highway.weight"] = uniform_sym(seed * 1000 + 601, (F, F), a)
    sd["highway.bias"] = uniform_sym(seed * 1000 + 602, (F,), a)
    sd["feature2out.weight"] = uniform_sym(seed * 1000 + 603, (n_out, F), a)
    sd["feature2out.bias"] = uniform_sym(seed * 1000 + 604, (n_out,), a)
    return sd
