"""Multi-GPU helpers: one process per GPU, knot ranges sharded across ranks.

Every interval is independent given Z, and each rank's Jacobian / Hessian / gradient output is ONE
contiguous slab of the global value vector (the z_{k+1} halves of the integrator blocks are constant,
so the owner of knot k emits all of column block k).  The data path therefore needs no collective;
`allgather_slabs` (RCCL all-gather over xGMI, or gloo on CPU) is only for consumers that want the
whole vector on every rank, and `allreduce_sum` is for the objective's partial sums."""
from __future__ import annotations


def shard_ranges(N, world, cost=None):
    """Contiguous 1-based inclusive knot ranges [(k_lo, k_hi)] for `world` ranks: balanced by knot count, or -- given `cost`, the
    per-interval cost model of the engine (Evaluator.interval_costs / dto_interval_costs, length N - 1: interval k belongs to knot
    k) -- by cost: the contiguous partition with the smallest maximum rank cost ("balanced by sum s_k if
    scaling counts vary").  Cost-balanced ranges are unequal in knots, so their slabs travel in the broadcast form of the gather."""
    if cost is None:
        base, rem = divmod(N, world)
        out, lo = [], 1
        for r in range(world):
            n = base + (1 if r < rem else 0)
            out.append((lo, lo + n - 1))
            lo += n
        return out
    import numpy as np
    c = np.concatenate([np.asarray(cost, dtype=np.float64), [0.0]])  # the last knot owns no interval
    if c.size != N or world > N or not np.all(np.isfinite(c)) or np.any(c < 0):
        raise ValueError("shard_ranges: cost must hold N - 1 finite non-negative entries and world <= N")
    pre = np.concatenate([[0.0], np.cumsum(c)])

    def parts(limit):
        """greedy packing under `limit` per rank, every rank at least one knot and enough knots left for the ranks behind it"""
        cuts, lo = [], 0
        for r in range(world):
            left = world - r - 1
            hi = int(np.searchsorted(pre, pre[lo] + limit, side="right")) - 1   # knots lo .. hi-1 cost <= limit
            hi = max(lo + 1, min(hi, N - left))
            if r == world - 1:
                hi = N
            cuts.append((lo, hi))
            lo = hi
        return cuts, max(pre[b] - pre[a] for a, b in cuts)

    lo_l, hi_l = max(float(c.max()), pre[-1] / world), float(pre[-1])
    best = parts(hi_l)
    for _ in range(60):   # bisection on the bottleneck cost
        mid = 0.5 * (lo_l + hi_l)
        cand = parts(mid)
        if cand[1] <= mid * (1 + 1e-12):
            best, hi_l = cand, mid
        else:
            lo_l = mid
    return [(a + 1, b) for a, b in best[0]]


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


def slab_layout(lo, n, group=None):
    """Every rank's (offset, length) of one value vector, from each rank's dto_shard_info entry."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    pairs = [None] * world
    dist.all_gather_object(pairs, (int(lo), int(n)), group=group)
    return pairs


def padded_gather_plan(layout):
    """One equal-size all-gather moves knot-range slabs IN PLACE if the vector is allocated with a little padding: the slabs
    of the interior ranks are equally long (n) and back to back, only the first and the last are shorter (knot 1 has no
    z_{k+1} half, knot N no own half).  With n - n_first doubles in front and n - n_last behind, rank r's slab lies inside
    chunk r of a buffer of world * n doubles (at the chunk's end for rank 0, at its start for the others).
    Returns (n, pad_front, pad_back) or None when the layout does not have that shape."""
    world = len(layout)
    n = max(ln for _, ln in layout)
    lo0, n0 = layout[0]
    if n == 0:
        return None
    for r in range(1, world):
        lo, ln = layout[r]
        if lo != lo0 + n0 + (r - 1) * n or ln > n or (r < world - 1 and ln != n):
            return None
    return n, n - n0, n - layout[-1][1]


def alloc_gather_vector(total, layout, dtype, device):
    """(buffer, full): `full` is the value vector of `total` entries every rank's engine writes its slab into
    (full[lo_r : lo_r + n_r]); `buffer` is the padded allocation behind it that `gather_slabs_inplace` hands to the
    equal-size all-gather (None when the layout needs the per-rank broadcasts)."""
    import torch
    plan = padded_gather_plan(layout)
    if plan is None or sum(ln for _, ln in layout) != total or layout[0][0] != 0:
        return None, torch.empty(total, dtype=dtype, device=device)
    n, front, back = plan
    buf = torch.empty(front + total + back, dtype=dtype, device=device)
    buf[:front].zero_()
    buf[front + total:].zero_()
    return buf, buf[front:front + total]


def gather_slabs_inplace(full, layout, group=None, buffer=None):
    """All-gather of knot-range slabs WITHOUT staging copies: `full` is the whole value vector (allocated on every rank),
    rank r's engine has written its slab straight into full[lo_r : lo_r + n_r] (the `*_dev` entry points take the slice's
    pointer), and this call fills in the other ranks' slices.

    With the padded allocation of `alloc_gather_vector` (`buffer`) that is ONE all_gather_into_tensor on the buffer itself --
    every link busy at once, the 2.2 GB slabs of configs[3] never copied.  Without it: the same single call when the slabs
    happen to be equal and back to back, else one in-place broadcast per rank (same bytes, but one source at a time).
    Backend "nccl" is RCCL over xGMI on the GPU box; "gloo" serves the CPU and one-device rehearsals."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(layout) == world
    one_call = dist.get_backend(group) == "nccl" or not full.is_cuda
    if buffer is not None and one_call:
        n, front, back = padded_gather_plan(layout)
        assert buffer.numel() == world * n and full.data_ptr() == buffer.data_ptr() + front * buffer.element_size()
        dist.all_gather_into_tensor(buffer, buffer[rank * n:(rank + 1) * n], group=group)
        return full
    lens = {n for _, n in layout}
    contiguous = all(layout[r][0] == layout[0][0] + r * layout[0][1] for r in range(world))
    if len(lens) == 1 and contiguous and one_call:
        lo0, n = layout[0]
        dist.all_gather_into_tensor(full[lo0:lo0 + world * n], full[layout[rank][0]:layout[rank][0] + n], group=group)
        return full
    for r, (lo, n) in enumerate(layout):
        if n:
            dist.broadcast(full[lo:lo + n], src=dist.get_global_rank(group, r) if group is not None else r, group=group)
    return full


def gather_slabs_async(full, layout, group=None):
    """The in-place gather as asynchronous collectives: one broadcast per rank, enqueued behind the work that is on the
    current stream NOW (the rank's own slab), running on the backend's own stream next to whatever is enqueued afterwards.
    Returns the work handles; `wait()` on each makes the current stream wait for it.

    This is how the gather overlaps with compute: a rank splits its knot range over two or more
    engine handles (`split_range`), and while handle i+1 computes, the slabs of handle i are already on the links."""
    import torch.distributed as dist
    works = []
    for r, (lo, n) in enumerate(layout):
        if n:
            src = dist.get_global_rank(group, r) if group is not None else r
            works.append(dist.broadcast(full[lo:lo + n], src=src, group=group, async_op=True))
    return works


def split_range(k_lo, k_hi, parts):
    """Contiguous, balanced 1-based inclusive sub-ranges of [k_lo, k_hi] (a rank's knots over `parts` handles)."""
    return [(k_lo + a - 1, k_lo + b - 1) for a, b in shard_ranges(k_hi - k_lo + 1, parts)]


def allreduce_sum(t, group=None):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
