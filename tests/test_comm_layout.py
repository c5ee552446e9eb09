"""CPU: the multi-GPU bookkeeping behind the C ABI (include/dto_engine.h "Multi-GPU", csrc/dto_comm.*) on structure-only
handles -- every rank's slabs from the knot ranges alone, the padded allocation that lets ONE in-place all-gather move
unequal slabs, the broadcast fallback for other assignments, the row segments of g.  The collectives themselves need RCCL
and GPUs (tests/test_gpu_comm.py); the Python twin of the same layout logic (host/distributed.py) is checked against it."""
import numpy as np
import pytest

import dto_amd
import dto_oracle as O
from dto_amd import capi
from helpers import to_engine


def _handles(p, ranges):
    evs = []
    for r, (lo, hi) in enumerate(ranges):
        ev = dto_amd.Evaluator(to_engine(p), device=-1, k_lo=lo, k_hi=hi)
        ev.comm_set_ranges(r, ranges)
        evs.append(ev)
    return evs


@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_layouts_agree_with_shard_info_and_the_python_twin(world):
    p = O.make_scaled_problem(11, 4, 2, seed=6, with_constraint=True)
    ranges = dto_amd.distributed.shard_ranges(p.N, world)
    evs = _handles(p, ranges)
    for vec, lo_f, len_f, total in ((capi.VECTOR_JACOBIAN, "jac_lo", "jac_len", evs[0].n_jacobian_entries),
                                    (capi.VECTOR_HESSIAN, "hess_lo", "hess_len", evs[0].n_hessian_entries),
                                    (capi.VECTOR_GRADIENT, "grad_lo", "grad_len", evs[0].n_variables)):
        want = [(getattr(e.shard, lo_f), getattr(e.shard, len_f)) for e in evs]
        for r, e in enumerate(evs):
            assert e.gather_slabs(vec) == want                    # every rank derives every rank's slab by itself
            L = e.gather_layout(vec)
            assert (L.total, L.own_lo, L.own_len, L.world) == (total, want[r][0], want[r][1], world)
            plan = dto_amd.distributed.padded_gather_plan(want)
            assert plan is not None and L.in_place_all_gather == 1
            n, front, back = plan
            assert (L.front_pad, L.padded_len) == (front, front + total + back) and L.padded_len == world * n
            # rank r's slab lies inside chunk r of the padded allocation: what the in-place all-gather needs
            a = L.front_pad + L.own_lo
            assert r * n <= a and a + L.own_len <= (r + 1) * n
            # ... at the chunk's END for rank 0 and at its START for the others (back to back with the neighbours)
            assert (a + L.own_len == n) if r == 0 else (a == r * n)
    # simulate the in-place all-gather with numpy: chunk r of every rank's buffer := chunk r of rank r's buffer
    L = evs[0].gather_layout(capi.VECTOR_JACOBIAN)
    n = L.padded_len // world
    full = np.arange(L.total, dtype=np.float64)
    bufs = []
    for e in evs:
        Lr = e.gather_layout(capi.VECTOR_JACOBIAN)
        b = np.full(Lr.padded_len, np.nan)
        b[Lr.front_pad + Lr.own_lo:Lr.front_pad + Lr.own_lo + Lr.own_len] = full[Lr.own_lo:Lr.own_lo + Lr.own_len]
        bufs.append(b)
    out = np.concatenate([bufs[r][r * n:(r + 1) * n] for r in range(world)])
    assert np.array_equal(out[L.front_pad:L.front_pad + L.total], full)


def test_two_handles_per_rank_take_the_broadcast_form():
    # a rank's knots over two handles (the overlapped gather): handle set A = first halves, set B = second halves;
    # neither set tiles the vector, so the layout is the plain vector and the gather one broadcast per rank
    p = O.make_scaled_problem(12, 3, 1, seed=2)
    halves = [dto_amd.distributed.split_range(lo, hi, 2) for lo, hi in dto_amd.distributed.shard_ranges(p.N, 2)]
    for part in (0, 1):
        ranges = [h[part] for h in halves]
        evs = _handles(p, ranges)
        for r, e in enumerate(evs):
            L = e.gather_layout(capi.VECTOR_JACOBIAN)
            assert L.in_place_all_gather == 0 and L.front_pad == 0 and L.padded_len == L.total == e.n_jacobian_entries
            assert (L.own_lo, L.own_len) == (e.shard.jac_lo, e.shard.jac_len)
    # the four handles together cover the vector exactly once
    cover = np.zeros(evs[0].n_jacobian_entries, dtype=int)
    for part in (0, 1):
        for e in _handles(p, [h[part] for h in halves]):
            cover[e.shard.jac_lo:e.shard.jac_lo + e.shard.jac_len] += 1
    assert (cover == 1).all()


def test_constraint_rows_of_all_ranks_cover_g_once():
    p = O.make_scaled_problem(10, 4, 2, seed=3, with_constraint=True)
    ranges = dto_amd.distributed.shard_ranges(p.N, 3)
    evs = _handles(p, ranges)
    cover = np.zeros(evs[0].n_constraints, dtype=int)
    for e in evs:
        L = e.gather_layout(capi.VECTOR_CONSTRAINT)
        assert (L.total, L.padded_len, L.front_pad, L.own_len) == (e.n_constraints, e.n_constraints, 0, e.shard.cons_len)
        st, ln = e.shard_rows()
        assert ln.sum() == e.shard.cons_len
        for a, b in zip(st, ln):
            cover[a - 1:a - 1 + b] += 1
    assert (cover == 1).all()


def test_errors_are_reported_not_swallowed():
    p = O.make_scaled_problem(8, 3, 1, seed=1)
    ev = dto_amd.Evaluator(to_engine(p), device=-1, k_lo=1, k_hi=4)
    with pytest.raises(dto_amd.EngineError, match="no knot ranges"):
        ev.gather_layout(capi.VECTOR_JACOBIAN)
    with pytest.raises(dto_amd.EngineError, match="not the handle's own shard"):
        ev.comm_set_ranges(0, [(1, 5), (6, 8)])
    with pytest.raises(dto_amd.EngineError, match="out of bounds"):
        ev.comm_set_ranges(0, [(1, 4), (5, 9)])
    ev.comm_set_ranges(0, [(1, 4), (5, 8)])
    with pytest.raises(dto_amd.EngineError, match="unknown vector"):
        ev.gather_layout(9)
    with pytest.raises(dto_amd.EngineError, match="structure-only"):
        ev.gather_dev(capi.VECTOR_JACOBIAN, 0)          # collectives need a GPU and a communicator
