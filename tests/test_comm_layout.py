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


@pytest.mark.parametrize("world,N", [(1, 11), (2, 11), (3, 11), (4, 11), (8, 16), (8, 24)])
def test_layouts_agree_with_shard_info_and_the_python_twin(world, N):
    p = O.make_scaled_problem(N, 4, 2, seed=6, with_constraint=True)
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


def test_world_8_with_unequal_interior_ranks_takes_the_broadcast_form():
    # 19 knots over 8 ranks: three ranks of 3 knots, five of 2 -- interior slabs of different lengths cannot be moved by one
    # equal-size all-gather; the layout falls back to one in-place broadcast per rank on the plain vector
    p = O.make_scaled_problem(19, 3, 1, seed=4)
    ranges = dto_amd.distributed.shard_ranges(p.N, 8)
    assert [hi - lo + 1 for lo, hi in ranges] == [3, 3, 3, 2, 2, 2, 2, 2]
    evs = _handles(p, ranges)
    cover = np.zeros(evs[0].n_jacobian_entries, dtype=int)
    for e in evs:
        L = e.gather_layout(capi.VECTOR_JACOBIAN)
        assert L.world == 8 and L.in_place_all_gather == 0 and L.front_pad == 0 and L.padded_len == L.total
        cover[L.own_lo:L.own_lo + L.own_len] += 1
    assert (cover == 1).all()


@pytest.mark.parametrize("shape", ["configs[3]", "configs[4]"])
def test_world_8_layouts_of_the_baseline_configurations(shape):
    """The two 8-GPU configurations of BASELINE.json as the driver's scaling run will lay them out (structure-only handles, no GPU):
    configs[3] 256-state x 16000 knots, configs[4] 1024-state + inequality + L1 slack (z = 1037) x 4000 knots -- eight ranks, the
    padded value vector (front pad, eight equal chunks), every rank's slab inside its chunk, all slabs tiling the vector."""
    syn = dto_amd.host.synthetic
    if shape == "configs[3]":
        prob, N, z, n, m = syn.make_scaled_problem(16000, 256, 4, seed=42), 16000, 265, 256, 4
        jac_nnz = 2_204_662_200                                   # SURVEY.md section 8a, S1 (bilinear + derivative rows)
    else:
        prob, N, z, n, m = syn.make_l1_slack_problem(4000, 1024, 4, seed=42), 4000, 1037, 1024, 4
        jac_nnz = (N - 1) * (n + m) * 2 * z + (N - 2) * m          # integrator blocks + one norm row over u per listed knot
    world = 8
    ranges = dto_amd.distributed.shard_ranges(N, world)
    evs = []
    for r, (lo, hi) in enumerate(ranges):
        ev = dto_amd.Evaluator(prob, device=-1, k_lo=lo, k_hi=hi)
        ev.comm_set_ranges(r, ranges)
        evs.append(ev)
    assert evs[0].n_variables == N * z and evs[0].n_jacobian_entries == jac_nnz
    for vec, total in ((capi.VECTOR_JACOBIAN, evs[0].n_jacobian_entries), (capi.VECTOR_HESSIAN, evs[0].n_hessian_entries)):
        end = 0
        Ls = [e.gather_layout(vec) for e in evs]
        n_chunk = Ls[0].padded_len // world
        for r, L in enumerate(Ls):
            assert (L.world, L.total, L.in_place_all_gather) == (world, total, 1) and L.padded_len == world * n_chunk
            assert L.front_pad == n_chunk - Ls[0].own_len                # rank 0's slab ends where chunk 1 begins
            a = L.front_pad + L.own_lo
            assert r * n_chunk <= a and a + L.own_len <= (r + 1) * n_chunk
            assert L.own_lo == end                                       # the slabs tile the vector in rank order
            end += L.own_len
        assert end == total
        # interior ranks are equally long; the padding is a boundary half-block, not a chunk
        assert len({L.own_len for L in Ls[1:-1]}) == 1 and Ls[0].padded_len - total < 2 * (Ls[1].own_len - min(Ls[0].own_len, Ls[-1].own_len)) + 1
    for e in evs:
        e.close()


def test_cost_balanced_knot_ranges():
    """"balanced by sum s_k if scaling counts vary".  A pulse whose amplitude (and time step) grows along the
    trajectory makes late intervals several squarings and Taylor terms dearer than early ones: equal knot counts leave the last rank
    with far more than its share, ranges cut by the engine's own cost model (dto_interval_costs, host arithmetic, also on a
    structure-only handle) bring every rank within 5 % of the mean."""
    N, n, m, world = 1601, 48, 2, 8
    p = O.make_scaled_problem(N, n, m, seed=5)
    Z = p.Z0.copy().reshape(N, p.z)
    ramp = np.linspace(0.05, 4.0, N)
    Z[:, n:n + m] *= 30.0 * ramp[:, None]           # drive amplitude grows by two orders of magnitude
    Z[:, p.dt_idx] = 0.02 + 0.2 * ramp / 4.0        # and so does the time step
    Z = Z.ravel()
    ev = dto_amd.Evaluator(to_engine(p), device=-1)
    cost = ev.interval_costs(Z)
    assert cost.shape == (N - 1,) and np.all(cost > 0) and cost[-1] > 1.5 * cost[0]
    part = ev.interval_costs(Z, first=100, count=50)
    assert np.array_equal(part, cost[100:150])
    def rank_costs(ranges):
        return np.array([cost[lo - 1:min(hi, N - 1)].sum() for lo, hi in ranges])
    even = rank_costs(dto_amd.distributed.shard_ranges(N, world))
    assert even.max() > 1.15 * even.mean()          # the imbalance the knot-count split leaves
    ranges = dto_amd.distributed.shard_ranges(N, world, cost=cost)
    assert ranges[0][0] == 1 and ranges[-1][1] == N and all(ranges[r][1] + 1 == ranges[r + 1][0] for r in range(world - 1))
    bal = rank_costs(ranges)
    assert bal.max() <= 1.05 * bal.mean(), (bal / bal.mean()).round(3)
    # unequal ranges are served by the broadcast form of the gather
    evs = []
    for r, (lo, hi) in enumerate(ranges):
        e = dto_amd.Evaluator(to_engine(p), device=-1, k_lo=lo, k_hi=hi)
        e.comm_set_ranges(r, ranges)
        evs.append(e)
    cover = np.zeros(evs[0].n_jacobian_entries, dtype=int)
    for e in evs:
        L = e.gather_layout(capi.VECTOR_JACOBIAN)
        assert L.in_place_all_gather == 0
        cover[L.own_lo:L.own_lo + L.own_len] += 1
    assert (cover == 1).all()
    # degenerate inputs
    assert dto_amd.distributed.shard_ranges(5, 5, cost=np.ones(4)) == [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5)]
    with pytest.raises(ValueError):
        dto_amd.distributed.shard_ranges(5, 2, cost=np.ones(3))
