"""GPU parity of the 33..64-state forms (round 4): the one-launch propagator chain (csrc/dto_chain64.hip, option "chain_form") and
the generator-stationary K-split sweep (k_sweep_s64 in csrc/dto_sweep_fused.hip, option "sweep_form") against the oracle and
against the general forms they replace -- forms of both evaluation orders and several squarings in one launch, three and five
generator slots, single-interval and ragged tile fills, sub-stepped sweeps, non-finite iterates, shards."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, run_all, to_engine

pytestmark = pytest.mark.gpu


def _all(p, Z, options=(), hessian=True, **kw):
    import dto_amd
    ev = dto_amd.Evaluator(to_engine(p), eval_hessian=hessian, **kw)
    try:
        for k, v in options:
            ev.set_option(k, v)
        mu = np.random.default_rng(7).standard_normal(ev.n_constraints)
        return run_all(ev, p, Z, mu, sigma=0.7, hessian=hessian), mu
    finally:
        ev.close()


def _oracle(p, Z, mu, hessian=True):
    ev_o = O.OracleEvaluator(p)
    ref = {"cons": ev_o.eval_constraint(Z), "jac": ev_o.eval_constraint_jacobian(Z)}
    if hessian:
        ref["hess"] = ev_o.eval_hessian_lagrangian(Z, 0.7, mu)
    return ref


@pytest.mark.parametrize("n,m,N", [(33, 2, 7), (40, 4, 23), (64, 4, 50), (48, 1, 3), (57, 3, 2)])
def test_forms_against_the_oracle_and_the_general_forms(n, m, N):
    """k_chain64 + k_sweep_s64 (defaults) vs the oracle, and vs the batched-GEMM chain / the other sweep forms (options): m + 1 = 2..5
    generator slots (the 3- and 5-slot instances), 1..49 intervals (one workgroup to several, ragged last tile)."""
    p = O.make_scaled_problem(N, n, m, seed=100 * n + m)
    Z = p.Z0.copy()
    out, mu = _all(p, Z)
    ref = _oracle(p, Z, mu)
    assert rel_err(out["cons"], ref["cons"]) <= 1e-10
    assert rel_err(out["jac"], ref["jac"]) <= 1e-10
    assert rel_err(out["hess"], ref["hess"]) <= 1e-8
    gen, _ = _all(p, Z, options=(("chain_form", 1), ("sweep_form", 1)))
    assert rel_err(out["jac"], gen["jac"]) <= 1e-11 and rel_err(out["cons"], gen["cons"]) <= 1e-11
    assert rel_err(out["hess"], gen["hess"]) <= 1e-9


def test_both_evaluation_forms_and_squarings_in_one_launch():
    """dt from 1e-6 to 2 along the trajectory: per interval the kernel picks the two-product or the three-product form and 0..many
    squarings; the sweeps need sub-steps (q > 1 rounds of k_sweep_s64).  ||exp|| grows to ~1e6: tolerance relative to max(1, |ref|)."""
    p = O.make_scaled_problem(9, 40, 3, seed=41)
    Z = p.Z0.copy()
    Z[p.dt_idx::p.z] = [1e-6, 1e-3, 0.05, 0.3, 0.8, 1.5, 2.0, 0.1, 0.5]
    out, mu = _all(p, Z)
    ref = _oracle(p, Z, mu)
    assert rel_err(out["cons"], ref["cons"]) <= 1e-9
    assert rel_err(out["jac"], ref["jac"]) <= 1e-9
    assert rel_err(out["hess"], ref["hess"]) <= 1e-7
    for form in (2, 3):   # either form forced for every interval
        o2, _ = _all(p, Z, options=(("expm_form", form),), hessian=False)
        assert rel_err(o2["jac"], ref["jac"]) <= 1e-9


def test_non_finite_iterate_propagates_without_error():
    """A NaN state entry: the reference returns NaNs, it does not throw -- the one-launch forms must end (bounded work) and
    leave the other intervals' entries finite."""
    import dto_amd
    p = O.make_scaled_problem(6, 40, 2, seed=3)
    ev = dto_amd.Evaluator(to_engine(p))
    try:
        Z = p.Z0.copy()
        Z[2 * p.z + 1] = np.nan   # a state entry of knot 3
        g = np.zeros(ev.n_constraints); ev.eval_constraint(g, Z)
        j = np.zeros(ev.n_jacobian_entries); ev.eval_constraint_jacobian(j, Z)
        assert np.isnan(g).any() and np.isfinite(g[:40]).all()
        assert np.isnan(j).any() and np.isfinite(j[:40]).all()
        Z = p.Z0.copy()
        Z[3 * p.z + 40] = np.inf  # a drive of knot 4: A_4 itself is not finite
        ev.eval_constraint(g, Z)
        ev.eval_constraint_jacobian(j, Z)
        assert not np.isfinite(g).all() and np.isfinite(g[:40]).all()
        Z = p.Z0.copy()          # and the handle is as good as new afterwards
        ev.eval_constraint_jacobian(j, Z)
        assert rel_err(j, O.OracleEvaluator(p).eval_constraint_jacobian(Z)) <= 1e-10
    finally:
        ev.close()


def test_sharded_handles_tile_the_whole_problem():
    """Three knot ranges of a 64-state problem: each shard's one-launch chain and sweeps see a different first interval and a
    different tile fill; their slabs side by side are the whole vectors."""
    import dto_amd
    p = O.make_scaled_problem(31, 64, 4, seed=64)
    Z = p.Z0.copy()
    whole, mu = _all(p, Z)
    ref = _oracle(p, Z, mu)
    assert rel_err(whole["jac"], ref["jac"]) <= 1e-10
    parts_j, parts_h, parts_c = [], [], []
    for lo, hi in dto_amd.distributed.shard_ranges(31, 3):
        ev = dto_amd.Evaluator(to_engine(p), k_lo=lo, k_hi=hi)
        try:
            j = np.zeros(ev.shard.jac_len); ev.eval_constraint_jacobian(j, Z)
            h = np.zeros(ev.shard.hess_len); ev.eval_hessian_lagrangian(h, Z, 0.7, mu)
            parts_j.append(j); parts_h.append(h)
        finally:
            ev.close()
    assert rel_err(np.concatenate(parts_j), ref["jac"]) <= 1e-10
    assert rel_err(np.concatenate(parts_h), ref["hess"]) <= 1e-8


def test_repeated_calls_are_bit_identical():
    """Fixed summation orders in both kernels (partials of the K split added in wavefront order, norms by integer maxima)."""
    import torch
    import dto_amd
    dev = torch.device("cuda", 0)
    p = O.make_scaled_problem(700, 50, 4, seed=5)
    ev = dto_amd.Evaluator(to_engine(p))
    try:
        Z = torch.from_numpy(p.Z0).to(dev)
        mu = torch.from_numpy(np.random.default_rng(2).standard_normal(ev.n_constraints)).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        runs = []
        for _ in range(3):
            j = torch.full((ev.shard.jac_len,), float("nan"), dtype=torch.float64, device=dev)
            g = torch.full((ev.n_constraints,), float("nan"), dtype=torch.float64, device=dev)
            ev.eval_jacobian_dev(Z.data_ptr(), j.data_ptr(), st)
            ev.eval_constraint_dev(Z.data_ptr(), g.data_ptr(), st)
            torch.cuda.synchronize()
            runs.append((j, g))
        assert bool(torch.isfinite(runs[0][0]).all())
        for j, g in runs[1:]:
            assert torch.equal(j, runs[0][0]) and torch.equal(g, runs[0][1])
        assert ev.last_stats()[1] > 0
    finally:
        ev.close()
