"""GPU: the device TimeDependentBilinearIntegrator (csrc/dto_tdb.hip; reference
src/integrators/time_dependent_bilinear_integrator.jl:60-244) for the parametrised generator family
G(u, t) = sum_j ubar_j (G_j + sum_c phi_c(t) H_cj).

Oracle: `O.TimeDependentBilinearIntegrator` in oracle/dto_oracle.py -- the reference's ODE (f!, :102-106) and control
interpolation (:85-92) restated, integrated with the fixed-step RK4 scheme the kernel's ABI states, derivatives by complex step /
Richardson-extrapolated differences.  tests/test_oracle_pinning.py ties that restatement to the ODE's solution (scipy DOP853 at
rtol 1e-12: fourth-order convergence, 1e-11 at 16 sub-steps on the reference's own carrier test) and to 40-digit arithmetic;
tests/golden/tdb_*.npz freeze its outputs (the engine is checked against those in test_gpu_golden_and_shards.py).  Bars:
1e-10 values / Jacobian, 1e-8 Hessian.  The reference itself integrates adaptively (Tsit5 at 1e-3 / 1e-6), so its own outputs
are only that close to any fixed scheme: parity against a running Julia reference is unpinned here as everywhere."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, to_engine

pytestmark = pytest.mark.gpu


def _problem(order, on_device, N=6, n=4, m=2, seed=5, substeps=16, mods=True):
    """Engine-side description of O.make_tdb_problem (device kernel or, on_device=False, the host-evaluated merge path)."""
    return to_engine(O.make_tdb_problem(N=N, n=n, m=m, order=order, seed=seed, substeps=substeps, n_mods=2 if mods else 0),
                     tdb_on_device=on_device)


def _all(ev, Z, mu, sigma=0.6):
    g = np.empty(ev.shard.cons_len); ev.eval_constraint(g, Z)
    j = np.empty(ev.shard.jac_len); ev.eval_constraint_jacobian(j, Z)
    h = np.empty(ev.shard.hess_len); ev.eval_hessian_lagrangian(h, Z, sigma, mu)
    return g, j, h


@pytest.mark.parametrize("order,n,m,N,substeps", [(0, 4, 2, 6, 16), (1, 4, 2, 6, 16), (1, 16, 2, 4, 8), (0, 24, 1, 3, 8)])
def test_device_propagator_matches_the_oracle(order, n, m, N, substeps):
    import dto_amd
    po = O.make_tdb_problem(N=N, n=n, m=m, order=order, seed=40 + n, substeps=substeps)
    ev_o = O.OracleEvaluator(po)
    ev = dto_amd.Evaluator(to_engine(po))
    try:
        r, c = ev.jacobian_structure()
        assert np.array_equal(r, ev_o.jacobian_structure1()[0]) and np.array_equal(c, ev_o.jacobian_structure1()[1])
        r, c = ev.hessian_lagrangian_structure()
        assert np.array_equal(r, ev_o.hessian_structure1()[0]) and np.array_equal(c, ev_o.hessian_structure1()[1])
        Z = po.Z0
        mu = np.random.default_rng(1).standard_normal(ev.n_constraints)
        g, j, h = _all(ev, Z, mu)
        errs = (rel_err(g, ev_o.eval_constraint(Z)), rel_err(j, ev_o.eval_constraint_jacobian(Z)),
                rel_err(h, ev_o.eval_hessian_lagrangian(Z, 0.6, mu)))
        print("tdb device vs oracle", order, n, errs)
        assert errs[0] <= 1e-10 and errs[1] <= 1e-10 and errs[2] <= 1e-8, errs
    finally:
        ev.close()


@pytest.mark.parametrize("order", [0, 1])
def test_host_evaluated_closure_path_matches_the_oracle(order):
    """The same integrator with its generator treated as an arbitrary closure: evaluated on the host, merged by the engine
    (DTO_INTEGRATOR_EXTERNAL) -- what a G(u, t) outside the device family takes."""
    import dto_amd
    po = O.make_tdb_problem(N=5, n=4, m=2, order=order, seed=9, substeps=16)
    ev_o = O.OracleEvaluator(po)
    ev = dto_amd.Evaluator(to_engine(po, tdb_on_device=False))
    try:
        Z = po.Z0
        mu = np.random.default_rng(2).standard_normal(ev.n_constraints)
        g, j, h = _all(ev, Z, mu)
        assert rel_err(g, ev_o.eval_constraint(Z)) <= 1e-10 and rel_err(j, ev_o.eval_constraint_jacobian(Z)) <= 1e-10
        assert rel_err(h, ev_o.eval_hessian_lagrangian(Z, 0.6, mu)) <= 1e-6   # the host path differentiates numerically
    finally:
        ev.close()


@pytest.mark.parametrize("order", [0, 1])
def test_device_propagator_matches_the_host_evaluated_path(order):
    import dto_amd
    pd, ph = _problem(order, True), _problem(order, False)
    ev_d, ev_h = dto_amd.Evaluator(pd), dto_amd.Evaluator(ph)
    try:
        for a, b in ((ev_d.jacobian_structure(), ev_h.jacobian_structure()),
                     (ev_d.hessian_lagrangian_structure(), ev_h.hessian_lagrangian_structure())):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        Z = pd.trajectory.vec()
        mu = np.random.default_rng(1).standard_normal(ev_d.n_constraints)
        gd, jd, hd = _all(ev_d, Z, mu)
        gh, jh, hh = _all(ev_h, Z, mu)
        errs = (rel_err(gd, gh), rel_err(jd, jh), rel_err(hd, hh))
        print("tdb device vs host mirror", order, errs)
        assert errs[0] <= 1e-10 and errs[1] <= 1e-10 and errs[2] <= 1e-8, errs
        # the blocks really depend on t and on u_{k+1} (order 1): structural zeros elsewhere
        r, c = ev_d.jacobian_structure()
        z = pd.trajectory.dim
        tcol = z - 2   # O.make_tdb_problem: components x, u, t, dt
        assert np.abs(jd[(c - 1) % z == tcol]).max() > 1e-3
    finally:
        ev_d.close(); ev_h.close()


def test_sharded_device_propagator_needs_no_halo_exchange():
    """Interval k's block has entries in the columns of knot k+1 (u_{k+1} with spline order 1): the owner of knot k+1
    evaluates interval k itself, so per-rank slabs still tile the full vectors."""
    import dto_amd
    po = O.make_tdb_problem(N=9, n=4, m=2, order=1, seed=5, substeps=16)
    p = to_engine(po)
    ev_o = O.OracleEvaluator(po)
    Z = po.Z0
    mu = np.random.default_rng(2).standard_normal(ev_o.n_constraints)
    g, j, h = ev_o.eval_constraint(Z), ev_o.eval_constraint_jacobian(Z), ev_o.eval_hessian_lagrangian(Z, 0.6, mu)
    for world in (2, 3):
        gj, gh = np.full_like(j, np.nan), np.full_like(h, np.nan)
        gg = np.full_like(g, np.nan)
        for lo, hi in dto_amd.distributed.shard_ranges(9, world):
            e = dto_amd.Evaluator(p, k_lo=lo, k_hi=hi)
            s = e.shard
            a, b, c = _all(e, Z, mu)
            gj[s.jac_lo:s.jac_lo + s.jac_len] = b
            gh[s.hess_lo:s.hess_lo + s.hess_len] = c
            st, ln = e.shard_rows()
            pos = 0
            for x, y in zip(st, ln):
                gg[x - 1:x - 1 + y] = a[pos:pos + y]
                pos += y
            e.close()
        assert rel_err(gj, j) <= 1e-10 and rel_err(gh, h) <= 1e-8 and rel_err(gg, g) <= 1e-10


def test_time_independent_family_reproduces_the_bilinear_integrator():
    """No modulation, controls held: the flow is exp(dt G(u)); RK4 with 200 sub-steps agrees with the device
    BilinearIntegrator's exponential to the scheme's O(h^4) error."""
    import dto_amd
    rng = np.random.default_rng(8)
    N, n, m = 5, 3, 1
    traj = dto_amd.NamedTrajectory({"x": rng.standard_normal((n, N)), "u": 0.3 * rng.standard_normal((m, N)),
                                    "t": np.linspace(0, 1, N)[None, :], "dt": np.full((1, N), 0.2)}, timestep="dt")
    G = 0.7 * rng.standard_normal((m + 1, n, n))
    tdb = dto_amd.TimeDependentBilinearIntegrator(dto_amd.ModulatedGenerators(G), "x", "u", "t", traj, spline_order=0, substeps=200)
    bil = dto_amd.BilinearIntegrator(G, "x", "u", traj)
    ev = dto_amd.Evaluator(dto_amd.DirectTrajOptProblem(traj, dto_amd.NullObjective(), [tdb, bil]))
    Z = traj.vec()
    g = np.empty(ev.n_constraints); ev.eval_constraint(g, Z)
    d = n * (N - 1)
    assert np.allclose(g[:d], g[d:], atol=1e-10)
    J = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(J, Z)
    r, c = ev.jacobian_structure()
    M = np.zeros((ev.n_constraints, ev.n_variables)); M[r - 1, c - 1] = J
    assert np.allclose(M[:d], M[d:], atol=1e-9)
    ev.close()


def test_unsupported_spline_order_is_refused():
    import dto_amd
    with pytest.raises(ValueError, match="Unsupported spline order"):
        p = _problem(0, True)
        dto_amd.TimeDependentBilinearIntegrator(p.integrators[0].G, "x", "u", "t", p.trajectory, spline_order=2)
