"""Multi-GPU helpers: one process per GPU, knot ranges sharded across ranks.

Every interval is independent given Z, and each rank's Jacobian / Hessian / gradient output is ONE
contiguous slab of the global value vector (the z_{k+1} halves of the integrator blocks are constant,
so the owner of knot k emits all of column block k).  The data path therefore needs no collective;
`allgather_slabs` (RCCL all-gather over xGMI, or gloo on CPU) is only for consumers that want the
whole vector on every rank, and `allreduce_sum` is for the objective's partial sums."""
from __future__ import annotations


def shard_ranges(N, world):
    """Contiguous, balanced 1-based inclusive knot ranges [(k_lo, k_hi)] for `world` ranks."""
    base, rem = divmod(N, world)
    out, lo = [], 1
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((lo, lo + n - 1))
        lo += n
    return out


def allgather_slabs(local, lens, group=None):
    """Concatenate per-rank slabs of lengths `lens` (known from dto_shard_info) on every rank.
    Slabs are padded to the longest so a single equal-size all-gather moves them."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert len(lens) == world
    mx = max(lens)
    buf = torch.zeros(mx, dtype=local.dtype, device=local.device)
    buf[:local.numel()] = local
    out = torch.empty(world * mx, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return torch.cat([out[r * mx:r * mx + lens[r]] for r in range(world)])


def allreduce_sum(t, group=None):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
