"""Multi-GPU sharding of the rollout index.

The R*L*B completions of one get_reward call are independent given (weights,
sentences, uniforms).  Rank r owns a contiguous range of rollout indices i,
i.e. the contiguous rows [i_lo*L, i_hi*L) of the raw reward matrix [R*L, B];
every i holds a full given_num = 1..L sweep, so ranks are load-balanced.  Per
adversarial step the weights are broadcast once (`sync_weights`: one flattened
broadcast from the source rank), per call the only exchange is one all-gather
of the raw rows (RCCL over xGMI when the process group's backend is "nccl");
every rank then applies the view(B,L,R).mean(-1) reduce locally.

Sharding is opt-in: a ROLLOUT shards only over the process group it was given
(`ROLLOUT.group`, or `WORLD` for the default group).  A process that
initialised torch.distributed for something else (a data-parallel trainer that
feeds different sentences to every rank) keeps the single-rank behaviour.
"""
import os

import torch
import torch.distributed as dist

#: value of ROLLOUT.group that selects the default process group
WORLD = "world"


def _resolve(group):
    """None for the default group, else the ProcessGroup itself."""
    return None if (isinstance(group, str) and group == WORLD) else group


def _host_staged(g, t):
    """With a gloo group the collectives here run on host copies of CUDA tensors; RCCL ("nccl") groups take the device
    tensors directly.  Why: in this image (torch 2.10 + ROCm 7.0) ProcessGroupGloo's own CUDA-tensor path stalls for
    good when a collective is entered while the calling stream still has work in flight.  Established in round 3 with
    tools/diag/gloo_cuda_repro.py (logs: profiles/r03_gloo_cuda_stall.txt): it needs no product code (torch matmuls in
    front of all_gather_into_tensor are enough), no second process (a one-rank group stalls too) and no broadcast in
    front; it does not happen when the device is idle at entry (torch.cuda.synchronize() first: 6 of 6 calls pass),
    and the same event / side-stream / pinned-copy sequence written by hand in two processes sharing the GPU passes.
    `.cpu()` below drains the stream and hands gloo host memory, which is the path gloo is built for."""
    if os.environ.get("TG_DIST_GLOO_DEVICE") == "1":      # diagnosis only (tools/diag/gloo_cuda_repro.py)
        return False
    return t.is_cuda and dist.get_backend(g) == "gloo"


def _all_gather_into(full, part, g):
    if _host_staged(g, part):
        host = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(host, part.cpu().contiguous(), group=g)
        full.copy_(host)
    else:
        dist.all_gather_into_tensor(full, part.contiguous(), group=g)


class ShardPlan:
    """Balanced contiguous split of range(R) over `world` ranks."""

    def __init__(self, R, world=1, rank=0):
        if not (0 <= rank < world):
            raise ValueError("rank %d outside world %d" % (rank, world))
        self.R, self.world, self.rank = R, world, rank
        q, r = divmod(R, world)
        self.counts = [q + (1 if k < r else 0) for k in range(world)]
        self.starts = [sum(self.counts[:k]) for k in range(world)]
        self.i_lo = self.starts[rank]
        self.i_hi = self.i_lo + self.counts[rank]
        self.max_count = max(self.counts) if world > 0 else 0

    @property
    def even(self):
        return self.R % self.world == 0

    @classmethod
    def from_group(cls, R, group=None):
        """group None: one rank, whatever torch.distributed's state is.  WORLD or a ProcessGroup: its ranks."""
        if group is None:
            return cls(R, 1, 0)
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("textgan_hip: ROLLOUT.group is set but torch.distributed is not initialised")
        g = _resolve(group)
        return cls(R, dist.get_world_size(g), dist.get_rank(g))


def gather_rows(raw_local, plan, L, group=None):
    """All-gather the per-rank raw reward rows into the full [R*L, B] matrix.

    raw_local: [count*L, B] rows of this rank (count = plan.counts[rank])."""
    if plan.world == 1:
        return raw_local
    g = _resolve(group)
    B = raw_local.shape[1]
    if plan.even:
        full = torch.empty(plan.R * L, B, dtype=raw_local.dtype, device=raw_local.device)
        _all_gather_into(full, raw_local, g)
        return full
    # uneven split: pad every rank to max_count rollouts, gather, drop the padding
    pad = torch.zeros(plan.max_count * L, B, dtype=raw_local.dtype, device=raw_local.device)
    pad[:raw_local.shape[0]] = raw_local
    full = torch.empty(plan.world * plan.max_count * L, B, dtype=raw_local.dtype, device=raw_local.device)
    _all_gather_into(full, pad, g)
    full = full.view(plan.world, plan.max_count * L, B)
    return torch.cat([full[k, :plan.counts[k] * L] for k in range(plan.world)], 0)


def _state_tensors(modules):
    seen, out = set(), []
    for m in modules:
        if m is None:
            continue
        for t in list(m.parameters()) + list(m.buffers()):
            if t.data_ptr() not in seen and t.numel():
                seen.add(t.data_ptr())
                out.append(t)
    return out


def sync_weights(gen, dis=None, group=WORLD, src=0):
    """The once-per-adversarial-step weight broadcast of SURVEY.md 8(e): every parameter and buffer of `gen` and
    `dis` on rank `src` of `group` (a rank *within the group*) replaces the other ranks' copies, in ONE flattened
    broadcast per dtype (a single one for the reference's all-fp32 modules).  The reference is single-process
    (instructor/oracle_data/seqgan_instructor.py:102-110 steps G between get_reward calls); with one process per GPU
    each running its own optimizer step -- and float atomics in the embedding gradient -- replicas drift by rounding,
    and the all-gathered raw rows would mix completions of slightly different generators.  Returns the number of
    bytes broadcast (0 on a single rank)."""
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return 0
    g = _resolve(group)
    if dist.get_world_size(g) == 1:
        return 0
    root = src if g is None else dist.get_global_rank(g, src)
    tensors = _state_tensors((gen, dis))
    nbytes = 0
    with torch.no_grad():
        by_kind = {}
        for t in tensors:
            by_kind.setdefault((t.dtype, t.device), []).append(t)
        for (dtype, device), ts in by_kind.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            if _host_staged(g, flat):
                host = flat.cpu()
                dist.broadcast(host, src=root, group=g)
                flat.copy_(host)
            else:
                dist.broadcast(flat, src=root, group=g)
            nbytes += flat.numel() * flat.element_size()
            if dist.get_rank(g) == src:
                continue                       # the source's own tensors already hold these values
            views, off = [], 0
            for t in ts:
                n = t.numel()
                views.append(flat[off:off + n].view_as(t))
                off += n
            torch._foreach_copy_([t.detach() for t in ts], views)   # one fused scatter back, not a launch per tensor
    return nbytes


def rank_seed(base, rank, stream=0):
    """A 63-bit seed for rank `rank`'s private random stream: splitmix64 of (base, rank, stream).  Ranks of a sharded
    get_reward call hold the same torch seed (same sentences, same weights), so drawing their uniforms from the
    default generator would make every rank's completions copies of rank 0's."""
    x = (int(base) + 0x9E3779B97F4A7C15 * (1 + int(rank) + (int(stream) << 32))) & 0xffffffffffffffff
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xffffffffffffffff
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xffffffffffffffff
    return int((x ^ (x >> 31)) & 0x7fffffffffffffff)
