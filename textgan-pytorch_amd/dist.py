"""Multi-GPU sharding of the rollout index.

The R*L*B completions of one get_reward call are independent given (weights,
sentences, uniforms).  Rank r owns a contiguous range of rollout indices i,
i.e. the contiguous rows [i_lo*L, i_hi*L) of the raw reward matrix [R*L, B];
every i holds a full given_num = 1..L sweep, so ranks are load-balanced.  The
only exchange is one all-gather of the raw rows (RCCL over xGMI when the
process group's backend is "nccl"); every rank then applies the
view(B,L,R).mean(-1) reduce locally.
"""
import torch
import torch.distributed as dist


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
        if dist.is_available() and dist.is_initialized():
            return cls(R, dist.get_world_size(group), dist.get_rank(group))
        return cls(R, 1, 0)


def gather_rows(raw_local, plan, L, group=None):
    """All-gather the per-rank raw reward rows into the full [R*L, B] matrix.

    raw_local: [count*L, B] rows of this rank (count = plan.counts[rank])."""
    if plan.world == 1:
        return raw_local
    B = raw_local.shape[1]
    if plan.even:
        full = torch.empty(plan.R * L, B, dtype=raw_local.dtype, device=raw_local.device)
        dist.all_gather_into_tensor(full, raw_local.contiguous(), group=group)
        return full
    # uneven split: pad every rank to max_count rollouts, gather, drop the padding
    pad = torch.zeros(plan.max_count * L, B, dtype=raw_local.dtype, device=raw_local.device)
    pad[:raw_local.shape[0]] = raw_local
    full = torch.empty(plan.world * plan.max_count * L, B, dtype=raw_local.dtype, device=raw_local.device)
    dist.all_gather_into_tensor(full, pad, group=group)
    full = full.view(plan.world, plan.max_count * L, B)
    return torch.cat([full[k, :plan.counts[k] * L] for k in range(plan.world)], 0)
