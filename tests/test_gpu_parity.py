"""GPU parity: every MOI callback of the HIP engine (through the C ABI) against the oracle.

Tolerances: values within 1e-10*max(1,|ref|) for objective, gradient, constraints
and Jacobian, 1e-8*max(1,|ref|) for the Hessian; sparsity indices bit-exact."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, run_all, to_engine

pytestmark = pytest.mark.gpu

TOL, TOL_H = 1e-10, 1e-8


def _check(prob_o, Z=None, seed=0, hessian=True, tag="", closure_derivatives="numeric", tol_h=TOL_H):
    import dto_amd
    ev_o = O.OracleEvaluator(prob_o)
    ev = dto_amd.Evaluator(to_engine(prob_o, closure_derivatives), eval_hessian=hessian)
    try:
        assert ev.n_variables == prob_o.n_vars
        assert ev.n_constraints == ev_o.n_constraints
        assert ev.n_dynamics_constraints == ev_o.n_dynamics_constraints
        jr, jc = ev.jacobian_structure()
        r1, c1 = ev_o.jacobian_structure1()
        assert np.array_equal(jr, r1) and np.array_equal(jc, c1), "Jacobian structure"
        hr, hc = ev.hessian_lagrangian_structure()
        r1, c1 = ev_o.hessian_structure1()
        assert np.array_equal(hr, r1) and np.array_equal(hc, c1), "Hessian structure"
        lo, hi = ev.constraint_bounds()
        lo_o, hi_o = ev_o.row_bounds()
        assert np.array_equal(lo, lo_o) and np.array_equal(hi, hi_o)
        rng = np.random.default_rng(seed)
        Z = prob_o.Z0.copy() if Z is None else Z
        mu = rng.standard_normal(ev_o.n_constraints)
        out = run_all(ev, prob_o, Z, mu, sigma=0.7, hessian=hessian)
        errs = {
            "f": rel_err(out["f"], ev_o.eval_objective(Z)),
            "grad": rel_err(out["grad"], ev_o.eval_objective_gradient(Z)),
            "cons": rel_err(out["cons"], ev_o.eval_constraint(Z)),
            "jac": rel_err(out["jac"], ev_o.eval_constraint_jacobian(Z)),
        }
        if hessian:
            errs["hess"] = rel_err(out["hess"], ev_o.eval_hessian_lagrangian(Z, 0.7, mu))
        print(tag, errs, ev.last_stats())
        for k, v in errs.items():
            assert v <= (tol_h if k == "hess" else TOL), (tag, k, v)
    finally:
        ev.close()


def test_readme_problem():
    _check(O.make_readme_problem(), tag="readme")


def test_standard_problem():
    _check(O.make_standard_problem(N=10), tag="standard")


def test_type1_derivative_only():
    _check(O.make_type1_derivative_problem(), tag="type1")


@pytest.mark.parametrize("n,m,N", [(3, 1, 4), (8, 2, 6), (17, 3, 5), (64, 4, 6), (70, 2, 4)])
def test_scaled_problems(n, m, N):
    _check(O.make_scaled_problem(N, n, m, seed=n + N, with_constraint=True), tag=f"scaled{n}x{N}")


def test_perturbed_point_and_large_norm():
    p = O.make_scaled_problem(5, 16, 2, seed=11, with_constraint=True)
    rng = np.random.default_rng(1)
    Z = p.Z0 + 0.3 * rng.standard_normal(p.n_vars)
    Z[p.dt_idx::p.z] = 0.9 + 0.2 * rng.random(p.N)  # big steps: ||dt G||_1 ~ 15, several squarings, q > 1
    _check(p, Z=Z, tag="largenorm")


def test_skew_generators_norm_preserved():
    _check(O.make_scaled_problem(6, 32, 2, seed=5, skew=True), tag="skew")


def test_jacobian_only_handle():
    _check(O.make_scaled_problem(5, 8, 2, seed=2), hessian=False, tag="nohess")


def test_closure_terms_merged_from_host_blocks():
    """closure-based knot constraint (2 outputs, repeated knots, between two built-in
    constraints) and knot objective (unsorted components) evaluated on the host and merged by the engine.  With the
    analytic derivatives handed through the bars are the usual ones; with the host mirror's numeric differentiation
    (complex step; differences for second derivatives) the Hessian bar is that of the differences."""
    p = O.make_closure_problem()
    _check(p, tag="closure/analytic", closure_derivatives="analytic")
    Z = p.Z0 + 0.05 * np.random.default_rng(3).standard_normal(p.n_vars)
    _check(p, Z=Z, tag="closure/analytic/perturbed", closure_derivatives="analytic")
    _check(p, Z=Z, tag="closure/numeric", closure_derivatives="numeric", tol_h=1e-6)


def test_closure_terms_sharded_and_products():
    import dto_amd
    p = O.make_closure_problem(N=10)
    ev_o = O.OracleEvaluator(p)
    Z = p.Z0 + 0.02 * np.random.default_rng(5).standard_normal(p.n_vars)
    mu = np.random.default_rng(6).standard_normal(ev_o.n_constraints)
    ref = {"jac": ev_o.eval_constraint_jacobian(Z), "hess": ev_o.eval_hessian_lagrangian(Z, 0.3, mu),
           "grad": ev_o.eval_objective_gradient(Z), "cons": ev_o.eval_constraint(Z)}
    got = {k: np.full_like(v, np.nan) for k, v in ref.items()}
    f = 0.0
    for lo, hi in dto_amd.distributed.shard_ranges(p.N, 3):
        ev = dto_amd.Evaluator(to_engine(p, "analytic"), k_lo=lo, k_hi=hi)
        s = ev.shard
        f += ev.eval_objective(Z)
        o = np.empty(s.grad_len); ev.eval_objective_gradient(o, Z); got["grad"][s.grad_lo:s.grad_lo + s.grad_len] = o
        o = np.empty(s.jac_len); ev.eval_constraint_jacobian(o, Z); got["jac"][s.jac_lo:s.jac_lo + s.jac_len] = o
        o = np.empty(s.hess_len); ev.eval_hessian_lagrangian(o, Z, 0.3, mu); got["hess"][s.hess_lo:s.hess_lo + s.hess_len] = o
        o = np.empty(s.cons_len); ev.eval_constraint(o, Z)
        st, ln = ev.shard_rows()
        pos = 0
        for a, b in zip(st, ln):
            got["cons"][a - 1:a - 1 + b] = o[pos:pos + b]
            pos += b
        ev.close()
    assert rel_err(f, ev_o.eval_objective(Z)) <= 1e-12
    for k in ref:
        assert rel_err(got[k], ref[k]) <= (1e-8 if k == "hess" else 1e-10), k
    # J w and J' w with external rows (materialised path)
    ev = dto_amd.Evaluator(to_engine(p, "analytic"))
    try:
        rng = np.random.default_rng(7)
        w = rng.standard_normal(p.n_vars)
        y = np.empty(ev_o.n_constraints); ev.eval_constraint_jacobian_product(y, Z, w)
        assert rel_err(y, ev_o.eval_constraint_jacobian_product(Z, w)) <= 1e-10
        w = rng.standard_normal(ev_o.n_constraints)
        y = np.empty(p.n_vars); ev.eval_constraint_jacobian_transpose_product(y, Z, w)
        assert rel_err(y, ev_o.eval_constraint_jacobian_transpose_product(Z, w)) <= 1e-10
        # a callback whose blocks were not supplied fails loudly instead of computing anything on the host
        vals = (dto_amd.capi.ExternalValues * 2)()
        assert ev._lib.dto_set_external(ev.handle, 2, vals) == 0
        out = np.empty(ev.n_constraints)
        Zc = np.ascontiguousarray(Z)
        rc = ev._lib.dto_eval_constraint(ev.handle, Zc.ctypes.data_as(dto_amd.capi.c_double_p), out.ctypes.data_as(dto_amd.capi.c_double_p))
        assert rc != 0 and b"not supplied" in ev._lib.dto_last_error(ev.handle)
    finally:
        ev.close()


def test_ket_infidelity_builtin_kind():
    """the coherent-fidelity loss |1 - ||A v||^2| (ConstantLowRankHVP shape, knot_hvp.jl:45-84) on
    the device: terminal + weighted interior listings, one knot twice, both signs of (1 - F).  The same problem
    with the loss written as a host closure must give the same answers through the merge path."""
    import dto_amd
    p = O.make_ket_problem()
    _check(p, tag="ket/builtin")
    Z = p.Z0 + 0.03 * np.random.default_rng(2).standard_normal(p.n_vars)
    _check(p, Z=Z, tag="ket/builtin/perturbed")
    q = O.make_ket_problem()
    q.objectives = [O.as_closure_objective(t) if t.kind == "knot_lowrank" else t for t in q.objectives]
    _check(q, Z=Z, tag="ket/closure", closure_derivatives="analytic")
    # front end: TerminalObjective("lowrank_infidelity", ...) with the factor from a goal state
    traj = to_engine(p).trajectory
    A = dto_amd.ket_fidelity_factor([0.6, 0.0, 0.0, 0.8])
    assert np.array_equal(A, O.ket_fidelity_factor([0.6, 0.0, 0.0, 0.8]))
    t = dto_amd.TerminalObjective("lowrank_infidelity", "c0", traj, Q=1.0, A=A)
    assert t.times.tolist() == [p.N] and t.comps.tolist() == [0, 1, 2, 3]


@pytest.mark.parametrize("n,m,N", [(128, 2, 3), (192, 3, 3), (256, 4, 3)])
def test_large_state_shapes_against_the_oracle(n, m, N):
    """The state sizes of BASELINE configs[2..3] (256) and their neighbours, all callbacks incl. the Hessian, on few
    knots so that the oracle (scipy expm/expm_frechet, block-triangular second-order terms) finishes in seconds:
    all three take the generator-subspace powers; 192 (not a multiple of 128) runs the chain on 64x64 tiles.  (The plain
    three-GEMM power chain is what the 64-state cases above use.)"""
    _check(O.make_scaled_problem(N, n, m, seed=11), tag=f"large-state n={n}")


@pytest.mark.parametrize("n,m", [(64, 4), (128, 2)])
def test_both_polynomial_forms_of_the_matrix_exponential(n, m):
    """Option "expm_form": the two-product degree-16 form (radius 0.78) and the three-product order-26 form (radius 2.82) of
    the propagator chain are each checked against the oracle on the same point (alpha ~ 4.5: 3 squarings against 1), and
    against each other; n = 64 takes the plain power chain, n = 128 the generator-subspace powers."""
    import dto_amd
    p = O.make_scaled_problem(4, n, m, seed=3)
    Z = p.Z0.copy()
    Z[p.dt_idx::p.z] = 0.1 * np.sqrt(256.0 / n) * 1.1
    ref = O.OracleEvaluator(p).eval_constraint_jacobian(Z)
    ev = dto_amd.Evaluator(to_engine(p), eval_hessian=False)
    try:
        got, squarings = {}, {}
        for form in (2, 3, 0):
            ev.set_option("expm_form", form)
            got[form] = np.full(ev.shard.jac_len, np.nan)
            ev.eval_constraint_jacobian(got[form], Z)
            squarings[form] = ev.last_stats()[0]
            assert rel_err(got[form], ref) <= TOL, (form, rel_err(got[form], ref))
        print("squarings", squarings, "form 2 vs 3", rel_err(got[2], got[3]))
        assert squarings[3] < squarings[2]
        assert rel_err(got[2], got[3]) <= 1e-12
        assert np.array_equal(got[0], got[3])  # by cost: three products + 1 squaring beat two products + 3
        with pytest.raises(Exception):
            ev.set_option("expm_form", 4)
    finally:
        ev.close()


@pytest.mark.parametrize("n,m,scale", [(64, 3, 0.15), (128, 2, 0.5), (40, 4, 0.6)])
def test_no_squaring_inside_the_radius(n, m, scale):
    """Small steps (alpha below the radius of the polynomial form in use): the last polynomial product is exp(A_k) itself and
    stores into the Jacobian slab -- no squaring launch.  scale = 0.15 stays inside the degree-16 radius (two products),
    0.5 and 0.6 are inside the order-26 radius only; a trajectory that mixes both kinds of interval is checked as well."""
    import dto_amd
    p = O.make_scaled_problem(5, n, m, seed=n + m)
    Z = p.Z0.copy()
    Z[p.dt_idx::p.z] = 0.1 * np.sqrt(256.0 / n) * scale
    ev = dto_amd.Evaluator(to_engine(p), eval_hessian=False)
    try:
        for form in (2, 3, 0):
            ev.set_option("expm_form", form)
            j = np.full(ev.shard.jac_len, np.nan)
            ev.eval_constraint_jacobian(j, Z)
            sq = ev.last_stats()[0]
            assert rel_err(j, O.OracleEvaluator(p).eval_constraint_jacobian(Z)) <= TOL, (form, sq)
            if form == 3 or (form == 2 and scale <= 0.15):
                assert sq == 0, (form, sq)
        Zm = Z.copy()
        Zm[p.dt_idx::p.z] = 0.1 * np.sqrt(256.0 / n) * np.array([0.1, 1.5, 0.1, 3.0, 0.1])
        ev.set_option("expm_form", 0)
        j = np.full(ev.shard.jac_len, np.nan)
        ev.eval_constraint_jacobian(j, Zm)
        assert rel_err(j, O.OracleEvaluator(p).eval_constraint_jacobian(Zm)) <= TOL
    finally:
        ev.close()


def test_external_integrator_merged_from_host_blocks():
    """an integrator evaluated outside the engine (the shape TimeDependentBilinearIntegrator has:
    both knot halves of the Jacobian block, cross part of the Hessian block) placed between built-in integrators."""
    p = O.make_external_integrator_problem()
    _check(p, tag="extint/analytic", closure_derivatives="analytic")
    Z = p.Z0 + 0.05 * np.random.default_rng(4).standard_normal(p.n_vars)
    _check(p, Z=Z, tag="extint/numeric", closure_derivatives="numeric", tol_h=1e-6)


def test_external_integrator_sharded():
    import dto_amd
    p = O.make_external_integrator_problem(N=9)
    ev_o = O.OracleEvaluator(p)
    Z = p.Z0 + 0.02 * np.random.default_rng(5).standard_normal(p.n_vars)
    mu = np.random.default_rng(6).standard_normal(ev_o.n_constraints)
    ref = {"jac": ev_o.eval_constraint_jacobian(Z), "hess": ev_o.eval_hessian_lagrangian(Z, 0.3, mu), "cons": ev_o.eval_constraint(Z)}
    got = {k: np.full_like(v, np.nan) for k, v in ref.items()}
    for lo, hi in dto_amd.distributed.shard_ranges(p.N, 4):
        ev = dto_amd.Evaluator(to_engine(p, "analytic"), k_lo=lo, k_hi=hi)
        s = ev.shard
        o = np.empty(s.jac_len); ev.eval_constraint_jacobian(o, Z); got["jac"][s.jac_lo:s.jac_lo + s.jac_len] = o
        o = np.empty(s.hess_len); ev.eval_hessian_lagrangian(o, Z, 0.3, mu); got["hess"][s.hess_lo:s.hess_lo + s.hess_len] = o
        o = np.empty(s.cons_len); ev.eval_constraint(o, Z)
        st, ln = ev.shard_rows()
        pos = 0
        for a, b in zip(st, ln):
            got["cons"][a - 1:a - 1 + b] = o[pos:pos + b]
            pos += b
        ev.close()
    for k in ref:
        assert rel_err(got[k], ref[k]) <= (1e-8 if k == "hess" else 1e-10), k


def test_time_dependent_bilinear_through_the_merge_path():
    """TimeDependentBilinearIntegrator (host mirror: fixed-step RK4) in a full problem: the engine's merged Jacobian
    must equal finite differences of its merged constraint values, and with a time-independent G and zero-order hold
    the rows must coincide with a device BilinearIntegrator on the same data."""
    import dto_amd
    rng = np.random.default_rng(3)
    N = 6
    traj = dto_amd.NamedTrajectory({"x": rng.standard_normal((2, N)), "u": 0.3 * rng.standard_normal((1, N)),
                                    "t": np.linspace(0.0, 1.0, N)[None, :], "dt": np.full((1, N), 0.2)}, timestep="dt")
    G0, G1 = np.array([[-0.1, 1.0], [-1.0, -0.1]]), np.array([[0.0, 1.0], [1.0, 0.0]])
    tdb = dto_amd.TimeDependentBilinearIntegrator(lambda u, t: G0 + u[0] * G1, "x", "u", "t", traj, spline_order=0, substeps=64)
    bil = dto_amd.BilinearIntegrator(np.stack([G0, G1]), "x", "u", traj)
    prob = dto_amd.DirectTrajOptProblem(traj, dto_amd.QuadraticRegularizer("u", traj, 1.0), [tdb, bil])
    ev = dto_amd.Evaluator(prob)
    try:
        Z = traj.vec()
        g = np.empty(ev.n_constraints); ev.eval_constraint(g, Z)
        d = 2 * (N - 1)
        assert np.allclose(g[:d], g[d:], atol=1e-9)
        J = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(J, Z)
        r, c = ev.jacobian_structure()
        M = np.zeros((ev.n_constraints, ev.n_variables)); M[r - 1, c - 1] = J
        assert np.allclose(M[:d], M[d:], atol=1e-7)
        eps = 1e-6
        for col in (0, 2, 3, 5, 8):
            e = np.zeros_like(Z); e[col] = eps
            gp = np.empty_like(g); gm = np.empty_like(g)
            ev.eval_constraint(gp, Z + e); ev.eval_constraint(gm, Z - e)
            assert np.allclose((gp - gm) / (2 * eps), M[:, col], atol=1e-6)
        mu = rng.standard_normal(ev.n_constraints)
        H = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(H, Z, 1.0, mu)
        assert np.isfinite(H).all()
    finally:
        ev.close()


def test_global_terms_merged_from_host_blocks():
    """SURVEY.md §8f rank 2, Global* kinds: GlobalObjective, GlobalKnotPointObjective (repeated knots: blocks accumulate)
    and a NonlinearGlobalConstraint between knot constraints; their Hessian entries in the global-variable columns
    form the tail of the CSC order, the constraint's Jacobian entries sit in the global columns."""
    p = O.make_global_problem()
    _check(p, tag="global/analytic", closure_derivatives="analytic")
    Z = p.Z0 + 0.05 * np.random.default_rng(8).standard_normal(p.n_vars)
    _check(p, Z=Z, tag="global/numeric", closure_derivatives="numeric", tol_h=1e-6)


def test_global_terms_sharded():
    import dto_amd
    p = O.make_global_problem(N=9)
    ev_o = O.OracleEvaluator(p)
    Z = p.Z0 + 0.02 * np.random.default_rng(5).standard_normal(p.n_vars)
    mu = np.random.default_rng(6).standard_normal(ev_o.n_constraints)
    ev = dto_amd.Evaluator(to_engine(p, "analytic"))  # J w / J' w reach the global-variable columns too
    try:
        w = np.random.default_rng(7).standard_normal(p.n_vars)
        y = np.empty(ev_o.n_constraints); ev.eval_constraint_jacobian_product(y, Z, w)
        assert rel_err(y, ev_o.eval_constraint_jacobian_product(Z, w)) <= 1e-10
        w = np.random.default_rng(8).standard_normal(ev_o.n_constraints)
        y = np.empty(p.n_vars); ev.eval_constraint_jacobian_transpose_product(y, Z, w)
        assert rel_err(y, ev_o.eval_constraint_jacobian_transpose_product(Z, w)) <= 1e-10
    finally:
        ev.close()
    ref = {"jac": ev_o.eval_constraint_jacobian(Z), "hess": ev_o.eval_hessian_lagrangian(Z, 0.3, mu),
           "grad": ev_o.eval_objective_gradient(Z), "cons": ev_o.eval_constraint(Z)}
    got = {k: np.full_like(v, np.nan) for k, v in ref.items()}
    f = 0.0
    for lo, hi in dto_amd.distributed.shard_ranges(p.N, 3):
        ev = dto_amd.Evaluator(to_engine(p, "analytic"), k_lo=lo, k_hi=hi)
        s = ev.shard
        f += ev.eval_objective(Z)
        o = np.empty(s.grad_len); ev.eval_objective_gradient(o, Z); got["grad"][s.grad_lo:s.grad_lo + s.grad_len] = o
        o = np.empty(s.jac_len); ev.eval_constraint_jacobian(o, Z); got["jac"][s.jac_lo:s.jac_lo + s.jac_len] = o
        o = np.empty(s.hess_len); ev.eval_hessian_lagrangian(o, Z, 0.3, mu); got["hess"][s.hess_lo:s.hess_lo + s.hess_len] = o
        o = np.empty(s.cons_len); ev.eval_constraint(o, Z)
        st, ln = ev.shard_rows()
        pos = 0
        for a, b in zip(st, ln):
            got["cons"][a - 1:a - 1 + b] = o[pos:pos + b]
            pos += b
        ev.close()
    assert rel_err(f, ev_o.eval_objective(Z)) <= 1e-12
    for k in ref:
        assert rel_err(got[k], ref[k]) <= (1e-8 if k == "hess" else 1e-10), k
