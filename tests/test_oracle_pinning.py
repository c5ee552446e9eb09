"""Pins the oracle (oracle/dto_oracle.py) WITHOUT the reference runtime (no julia here):
  * the one literal fixture of the reference's tests (named_trajectory_type_1,
    test/test_utils.jl:57-82) with the closed-form DerivativeIntegrator answers
    (derivative_integrator.jl:45, 68-116);
  * complex-step differentiation (first order) and 60-digit mpmath differentiation (second order)
    of x_{k+1} - exp(dt G(u)) x_k  (bilinear_integrator.jl:81);
  * invariants of the reference's Pauli-generator problem (test/test_utils.jl:121-145);
  * the reference's own finite-difference bars (evaluator.jl:752 atol=rtol=1e-6; :790 atol=1e-2);
  * the closed-form index layout of SURVEY.md §3.6 against the literal re-enactment of
    evaluator.jl:119-209."""
import numpy as np
import pytest
import scipy.linalg as sla

import dto_oracle as O


def test_type1_derivative_closed_form():
    p = O.make_type1_derivative_problem()
    data = O.NAMED_TRAJECTORY_TYPE_1
    a, da, dt = data[8:10], data[10:12], data[14]
    ev = O.OracleEvaluator(p)
    g = ev.eval_constraint(p.Z0)
    want = np.concatenate([a[:, k + 1] - a[:, k] - dt[k] * da[:, k] for k in range(4)])
    assert np.array_equal(g, want)
    # a literal spot value: knot 2 -> 3, component a_1:  0.959151 - (-0.243953) - 0.2*0.0240775
    assert g[2] == pytest.approx(0.959151 + 0.243953 - 0.2 * 0.0240775, abs=1e-15)
    rows, cols = ev.jacobian_structure1()
    vals = ev.eval_constraint_jacobian(p.Z0)
    assert len(vals) == 2 * 15 * 2 * 4  # 2 z D K
    J = np.zeros((8, 75))
    J[rows - 1, cols - 1] = vals
    for k in range(4):
        blk = J[2 * k:2 * k + 2, 15 * k:15 * k + 30]
        exp = np.zeros((2, 30))
        exp[:, 8:10] = -np.eye(2)
        exp[:, 10:12] = -dt[k] * np.eye(2)
        exp[:, 14] = -da[:, k]
        exp[:, 15 + 8:15 + 10] = np.eye(2)
        assert np.array_equal(blk, exp)
    mu = np.arange(1.0, 9.0)
    hr, hc = ev.hessian_structure1()
    H = ev.eval_hessian_lagrangian(p.Z0, 1.0, mu)
    nz = {(int(r), int(c)): v for r, c, v in zip(hr, hc, H) if v != 0}
    exp = {}
    for k in range(4):
        for i in range(2):
            exp[(15 * k + 11 + i, 15 * k + 15)] = -mu[2 * k + i]  # (da_i, dt) cross term, 1-based
    assert nz == exp


def test_csc_order_matches_closed_form():
    """column (k,j) stores, for each integrator, interval k-1 rows then interval k rows."""
    for (N, n, m) in [(5, 2, 1), (6, 3, 2), (4, 4, 2)]:
        p = O.make_scaled_problem(N, n, m, seed=1)
        ev = O.OracleEvaluator(p)
        rows, cols = ev.jacobian_structure1()
        z, K = p.z, N - 1
        dims = [n, m]
        offs = [0, n * K]
        er, ec = [], []
        for k in range(1, N + 1):
            for j in range(1, z + 1):
                for d, off in zip(dims, offs):
                    for kk in (k - 1, k):
                        if 1 <= kk <= K:
                            for r in range(1, d + 1):
                                er.append(off + (kk - 1) * d + r)
                                ec.append((k - 1) * z + j)
        assert np.array_equal(rows, er) and np.array_equal(cols, ec)
        assert len(rows) == 2 * z * sum(dims) * K
        hr, hc = ev.hessian_structure1()
        er, ec = [], []
        for k in range(1, N + 1):
            for j in range(1, z + 1):
                if k >= 2:
                    for i in range(1, z + 1):
                        er.append((k - 2) * z + i); ec.append((k - 1) * z + j)
                for i in range(1, j + 1):
                    er.append((k - 1) * z + i); ec.append((k - 1) * z + j)
        assert np.array_equal(hr, er) and np.array_equal(hc, ec)
        assert len(hr) == N * z * (z + 1) // 2 + K * z * z


def _f_complex(integ, prob, zk):
    n, m = integ.x_dim, integ.u_dim
    dt = zk[prob.dt_idx]
    x = zk[integ.x_off:integ.x_off + n]
    u = zk[integ.u_off:integ.u_off + m]
    Gu = integ.G[0] + np.tensordot(u, integ.G[1:], axes=(0, 0))
    return -sla.expm(dt * Gu) @ x


def test_bilinear_jacobian_complex_step():
    p = O.make_scaled_problem(3, 6, 3, seed=9)
    integ = p.integrators[0]
    zk = p.Z0[:p.z].copy()
    zk[p.dt_idx] = 0.37
    B = O.bilinear_block_jacobian(integ, p, zk)
    h = 1e-30
    for j in range(p.z):
        zc = zk.astype(complex)
        zc[j] += 1j * h
        col = _f_complex(integ, p, zc).imag / h
        assert np.max(np.abs(col - B[:, j])) <= 1e-13 * max(1.0, np.abs(col).max())


def test_bilinear_hessian_mpmath():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    rng = np.random.default_rng(4)
    n, m = 2, 2
    G = rng.standard_normal((m + 1, n, n))
    p = O.Problem(N=2, z=n + m + 1, dt_idx=n + m, integrators=[O.BilinearIntegrator(0, n, n, m, G)],
                  Z0=np.zeros(2 * (n + m + 1)))
    zk = np.concatenate([rng.standard_normal(n), 0.4 * rng.standard_normal(m), [0.3]])
    mu = rng.standard_normal(n)
    H = O.bilinear_block_hessian(p.integrators[0], p, zk, mu)
    Gm = [mp.matrix(G[j].tolist()) for j in range(m + 1)]

    def phi(*zz):
        x = mp.matrix(zz[:n])
        Gu = Gm[0] + sum((zz[n + j] * Gm[1 + j] for j in range(m)), mp.zeros(n))
        E = mp.expm(zz[n + m] * Gu)
        y = E * x
        return -sum(mp.mpf(mu[i]) * y[i] for i in range(n))

    z0 = [mp.mpf(v) for v in zk]
    nz = len(z0)
    for a in range(nz):
        for b in range(a, nz):
            order = [0] * nz
            order[a] += 1
            order[b] += 1
            ref = mp.diff(phi, tuple(z0), tuple(order))
            assert abs(float(ref) - H[a, b]) <= 1e-11 * max(1.0, abs(float(ref))), (a, b)
            assert H[a, b] == pytest.approx(H[b, a], abs=1e-13)


def test_pauli_problem_invariants():
    p = O.make_standard_problem(N=8)
    integ = p.integrators[0]
    for G in integ.G:
        assert np.array_equal(G, -G.T)  # skew-symmetric generators: exp is orthogonal
    for k in range(p.K):
        zk = p.Z0[k * p.z:(k + 1) * p.z]
        B = O.bilinear_block_jacobian(integ, p, zk)
        E = -B[:, :4]
        assert np.allclose(E.T @ E, np.eye(4), atol=1e-14)
    g = O.integrator_evaluate(integ, p, p.Z0)
    for k in range(p.K):
        xk = p.Z0[k * p.z:k * p.z + 4]
        xk1 = p.Z0[(k + 1) * p.z:(k + 1) * p.z + 4]
        assert np.linalg.norm(xk1 - g[4 * k:4 * k + 4]) == pytest.approx(np.linalg.norm(xk), rel=1e-13)


def test_reference_finite_difference_bars():
    """The reference's own acceptance tests of this path are FD comparisons: Jacobian atol=rtol=1e-6
    (evaluator.jl:752), Hessian atol=1e-2 (:790), gradient `≈`.  The oracle passes them with margin."""
    p = O.make_standard_problem(N=6, seed=5)
    ev = O.OracleEvaluator(p)
    Z = p.Z0.copy()
    nv = p.n_vars
    eps = 1e-6
    rows, cols = ev.jacobian_structure1()
    J = np.zeros((ev.n_constraints, nv))
    J[rows - 1, cols - 1] = ev.eval_constraint_jacobian(Z)
    Jfd = np.stack([(ev.eval_constraint(Z + eps * e) - ev.eval_constraint(Z - eps * e)) / (2 * eps)
                    for e in np.eye(nv)], axis=1)
    assert np.allclose(J, Jfd, atol=1e-6, rtol=1e-6)
    mu = np.random.default_rng(0).standard_normal(ev.n_constraints)

    def gradL(zv):
        M = np.zeros((ev.n_constraints, nv))
        M[rows - 1, cols - 1] = ev.eval_constraint_jacobian(zv)
        return ev.eval_objective_gradient(zv) + M.T @ mu

    hr, hc = ev.hessian_structure1()
    Hu = np.zeros((nv, nv))
    Hu[hr - 1, hc - 1] = ev.eval_hessian_lagrangian(Z, 1.0, mu)
    Hs = Hu + np.triu(Hu, 1).T
    Hfd = np.stack([(gradL(Z + eps * e) - gradL(Z - eps * e)) / (2 * eps) for e in np.eye(nv)], axis=1)
    assert np.allclose(Hs, Hfd, atol=1e-5)
    gfd = np.array([(ev.eval_objective(Z + eps * e) - ev.eval_objective(Z - eps * e)) / (2 * eps) for e in np.eye(nv)])
    assert np.allclose(gfd, ev.eval_objective_gradient(Z), atol=1e-8)


def test_quadratic_regularizer_follows_code_not_docstring():
    """regularizers.jl:86-87 weights by dt^2 (the docstring says dt); (v,dt) Hessian entries survive
    only when the timestep follows v in the knot (regularizers.jl:160 + evaluator.jl:637)."""
    N, z = 3, 3
    data = np.array([[1.0, 2.0, 3.0], [0.5, 0.5, 0.5], [4.0, 5.0, 6.0]])  # comps: v, dt, w
    Z = data.T.reshape(-1)
    base = dict(N=N, z=z, dt_idx=1, integrators=[O.DerivativeIntegrator(0, 1, 2)], Z0=Z)
    pv = O.Problem(objectives=[O.QuadraticRegularizer(0, 1, np.array([2.0]))], **base)
    pw = O.Problem(objectives=[O.QuadraticRegularizer(2, 1, np.array([2.0]))], **base)
    assert O.objective_value(pv, Z) == pytest.approx(sum(0.5 * 2.0 * (0.5 * v) ** 2 for v in (1, 2, 3)))
    for prob, v_idx, keeps in ((pv, 0, True), (pw, 2, False)):
        ev = O.OracleEvaluator(prob)
        hr, hc = ev.hessian_structure1()
        H = ev.eval_hessian_lagrangian(Z, 1.0, np.zeros(ev.n_constraints))
        d = {(int(r), int(c)): v for r, c, v in zip(hr, hc, H)}
        lo, hi = sorted((v_idx + 1, 2))
        assert (d[(lo, hi)] != 0.0) == keeps


def test_closure_terms_pass_the_reference_fd_bars():
    """Closure-based knot constraint / objective of the oracle (analytic derivatives standing in for ForwardDiff)
    against finite differences at the reference's own tolerances (evaluator.jl:752, :790)."""
    p = O.make_closure_problem(N=7)
    ev = O.OracleEvaluator(p)
    Z = p.Z0.copy()
    nv = p.n_vars
    eps = 1e-6
    rows, cols = ev.jacobian_structure1()
    J = np.zeros((ev.n_constraints, nv))
    J[rows - 1, cols - 1] = ev.eval_constraint_jacobian(Z)
    Jfd = np.stack([(ev.eval_constraint(Z + eps * e) - ev.eval_constraint(Z - eps * e)) / (2 * eps) for e in np.eye(nv)], axis=1)
    # a knot listed twice owns two row blocks; both must match
    assert np.allclose(J, Jfd, atol=1e-6, rtol=1e-6)
    gfd = np.array([(ev.eval_objective(Z + eps * e) - ev.eval_objective(Z - eps * e)) / (2 * eps) for e in np.eye(nv)])
    g = ev.eval_objective_gradient(Z)
    # gradient!/hessian! overwrite per listed time (knot_point_objectives.jl:198): at the knot listed twice the
    # reference's gradient is that of the LAST listing only, so it differs from the FD of the summed value there
    dup = [(4 - 1) * p.z + c for c in (1, 6, 3)]
    keep = np.setdiff1d(np.arange(nv), dup)
    assert np.allclose(gfd[keep], g[keep], atol=1e-7)
    assert not np.allclose(gfd[dup], g[dup], atol=1e-3)


def test_ket_infidelity_loss_is_the_fidelity():
    """|1 - ||A psi~||^2| with A = ket_fidelity_factor(goal) equals 1 - |<goal|psi>|^2 for normalised states, and
    the oracle's gradient/Hessian of the term match finite differences away from the kink."""
    rng = np.random.default_rng(4)
    g = rng.standard_normal(3) + 1j * rng.standard_normal(3)
    g /= np.linalg.norm(g)
    psi = rng.standard_normal(3) + 1j * rng.standard_normal(3)
    psi /= np.linalg.norm(psi)
    A = O.ket_fidelity_factor(np.concatenate([g.real, g.imag]))
    v = np.concatenate([psi.real, psi.imag])
    assert abs(abs(1 - np.sum((A @ v) ** 2)) - (1 - abs(np.vdot(g, psi)) ** 2)) < 1e-14
    p = O.make_ket_problem(N=6)
    ev = O.OracleEvaluator(p)
    Z = p.Z0.copy()
    eps = 1e-6
    nv = p.n_vars
    gfd = np.array([(ev.eval_objective(Z + eps * e) - ev.eval_objective(Z - eps * e)) / (2 * eps) for e in np.eye(nv)])
    gr = ev.eval_objective_gradient(Z)
    dup = [(3 - 1) * p.z + c for c in range(4)]  # knot 3 is listed twice: last listing only (knot_point_objectives.jl:198)
    keep = np.setdiff1d(np.arange(nv), dup)
    assert np.allclose(gfd[keep], gr[keep], atol=1e-7)


def test_external_integrator_passes_the_reference_fd_bars():
    """The closure integrator of the oracle (shape of TimeDependentBilinearIntegrator's blocks: both knot halves, cross
    Hessian part) against finite differences at the reference's tolerances (_integrators.jl:97-242: 1e-3/1e-5)."""
    p = O.make_external_integrator_problem(N=5)
    ev = O.OracleEvaluator(p)
    Z = p.Z0.copy()
    nv, eps = p.n_vars, 1e-6
    rows, cols = ev.jacobian_structure1()
    J = np.zeros((ev.n_constraints, nv))
    J[rows - 1, cols - 1] = ev.eval_constraint_jacobian(Z)
    Jfd = np.stack([(ev.eval_constraint(Z + eps * e) - ev.eval_constraint(Z - eps * e)) / (2 * eps) for e in np.eye(nv)], axis=1)
    assert np.allclose(J, Jfd, atol=1e-6, rtol=1e-6)
    mu = np.random.default_rng(0).standard_normal(ev.n_constraints)

    def gradL(zv):
        M = np.zeros((ev.n_constraints, nv))
        M[rows - 1, cols - 1] = ev.eval_constraint_jacobian(zv)
        return ev.eval_objective_gradient(zv) + M.T @ mu

    hr, hc = ev.hessian_structure1()
    Hu = np.zeros((nv, nv))
    Hu[hr - 1, hc - 1] = ev.eval_hessian_lagrangian(Z, 1.0, mu)
    Hs = Hu + np.triu(Hu, 1).T
    Hfd = np.stack([(gradL(Z + eps * e) - gradL(Z - eps * e)) / (2 * eps) for e in np.eye(nv)], axis=1)
    assert np.allclose(Hs, Hfd, atol=1e-5)
    assert np.abs(Hs[:p.z, p.z:2 * p.z]).max() > 1e-3  # the cross (z_k, z_{k+1}) part is really exercised


def test_time_dependent_bilinear_mirror_reduces_to_the_exponential():
    """Host mirror of TimeDependentBilinearIntegrator: with a time-independent G and zero-order hold its defect is the
    BilinearIntegrator's (x_{k+1} - exp(dt G(u)) x_k) up to the RK4 step error, and with the carrier example of the
    reference's docstring (time_dependent_bilinear_integrator.jl:52-55) it matches a fine-step reference solve."""
    import dto_amd
    import scipy.linalg as sla
    rng = np.random.default_rng(1)
    N = 4
    traj = dto_amd.NamedTrajectory({"x": rng.standard_normal((2, N)), "u": 0.3 * rng.standard_normal((1, N)),
                                    "t": np.linspace(0.0, 0.6, N)[None, :], "dt": np.full((1, N), 0.2)}, timestep="dt")
    G0, G1 = np.array([[-0.1, 1.0], [-1.0, -0.1]]), np.array([[0.0, 1.0], [1.0, 0.0]])
    Zk = traj.vec().reshape(N, traj.dim)
    B = dto_amd.TimeDependentBilinearIntegrator(lambda u, t: G0 + u[0] * G1, "x", "u", "t", traj, spline_order=0, substeps=64)
    vals, jac, _ = B.external_blocks(Zk, 1)
    for k in range(N - 1):
        want = Zk[k + 1, 0:2] - sla.expm(0.2 * (G0 + Zk[k, 2] * G1)) @ Zk[k, 0:2]
        assert np.allclose(vals[k], want, atol=1e-10)
        assert np.allclose(jac[k].T[:, 0:2], -sla.expm(0.2 * (G0 + Zk[k, 2] * G1)), atol=1e-9)  # d/dx_k = -E
    Gt = lambda u, t: G0 + u[0] * np.array([[0.0, np.cos(t)], [np.cos(t), 0.0]])
    B1 = dto_amd.TimeDependentBilinearIntegrator(Gt, "x", "u", "t", traj, spline_order=1, substeps=16)
    B2 = dto_amd.TimeDependentBilinearIntegrator(Gt, "x", "u", "t", traj, spline_order=1, substeps=256)
    assert np.allclose(B1.external_blocks(Zk, 0)[0], B2.external_blocks(Zk, 0)[0], atol=1e-7)


# ----------------------------------------------------------------------------------------------
# TimeDependentBilinearIntegrator (oracle restatement: O.TimeDependentBilinearIntegrator)
# ----------------------------------------------------------------------------------------------

def _interval(p, k):
    return np.concatenate([p.Z0[k * p.z:(k + 1) * p.z], p.Z0[(k + 1) * p.z:(k + 2) * p.z]])


def test_tdb_fixed_step_map_converges_to_the_ode_solution_on_the_reference_carrier_example():
    """The reference integrates dy/dtau = G(u(tau), t_k + tau dt_k) (y dt_k) adaptively (Tsit5, default reltol 1e-3 / abstol
    1e-6, time_dependent_bilinear_integrator.jl:123-128) on its own test closure G(a) + 0.1 cos(t) I (:262-266).  The oracle's
    fixed-step RK4 map of the same right-hand side converges at fourth order to scipy's DOP853 solution at rtol 1e-12, and at the
    sub-step counts the tests use it is orders of magnitude inside the reference's own integration tolerance."""
    p = O.make_tdb_reference_carrier_problem()
    it = p.integrators[0]
    for k in (0, 4, 8):
        zz = _interval(p, k)
        want = zz[p.z:p.z + 4] - O.tdb_flow_reference(it, zz)
        errs = []
        for s in (1, 2, 4, 8, 16):
            it_s = O.TimeDependentBilinearIntegrator(it.x_off, it.x_dim, it.u_off, it.u_dim, it.t_off, it.G, it.mods,
                                                     it.spline_order, s).bind(p.z, p.dt_idx)
            errs.append(np.abs(it_s.f(zz) - want).max())
        assert errs[2] <= 1e-8 and errs[4] <= 1e-11, errs          # 4 sub-steps already beat Tsit5's abstol by 100x
        for a, b in zip(errs[:3], errs[1:4]):
            assert 10.0 <= a / b <= 24.0, errs                        # fourth order: 16x per halving (before rounding)


@pytest.mark.parametrize("n,m,order", [(4, 2, 1), (8, 3, 0), (16, 2, 1)])
def test_tdb_convergence_across_sizes_and_spline_orders(n, m, order):
    p = O.make_tdb_problem(N=3, n=n, m=m, order=order, seed=31 + n, substeps=8)
    it = p.integrators[0]
    zz = _interval(p, 0)
    want = zz[p.z:p.z + n] - O.tdb_flow_reference(it, zz)
    errs = []
    for s in (8, 16, 32, 64):
        it_s = O.TimeDependentBilinearIntegrator(it.x_off, n, it.u_off, m, it.t_off, it.G, it.mods, order, s).bind(p.z, p.dt_idx)
        errs.append(np.abs(it_s.f(zz) - want).max() / max(1.0, np.abs(want).max()))
    assert errs[-1] <= 1e-9 and errs[0] / errs[1] >= 9.0 and errs[1] / errs[2] >= 9.0, errs
    # spline order 1 really reads u_{k+1}; order 0 does not (time_dependent_bilinear_integrator.jl:85-92)
    J = it.jac(zz)
    next_u = np.abs(J[:, p.z + it.u_off:p.z + it.u_off + m]).max()
    assert (next_u > 1e-4) if order == 1 else (next_u == 0.0)
    assert np.abs(J[:, it.t_off]).max() > 1e-4                         # the carrier terms make the block depend on t_k


def test_tdb_time_independent_family_is_the_matrix_exponential():
    """No carrier terms, controls held: y(1) = exp(dt G(u)) x_k, the BilinearIntegrator's flow (bilinear_integrator.jl:81)."""
    rng = np.random.default_rng(3)
    n, m, z = 3, 1, 6
    G = 0.7 * rng.standard_normal((m + 1, n, n))
    it = O.TimeDependentBilinearIntegrator(0, n, n, m, 4, G, [], 0, 200).bind(z, 5)
    zz = rng.standard_normal(2 * z)
    zz[5] = 0.2
    E = sla.expm(0.2 * (G[0] + zz[3] * G[1]))
    assert np.allclose(it.f(zz), zz[z:z + n] - E @ zz[:n], atol=1e-12)
    J = it.jac(zz)
    assert np.allclose(J[:, :n], -E, atol=1e-11) and np.allclose(J[:, z:z + n], np.eye(n), atol=1e-14)
    assert np.allclose(J[:, 3], -sla.expm_frechet(0.2 * (G[0] + zz[3] * G[1]), 0.2 * G[1], compute_expm=False) @ zz[:n], atol=1e-11)


def test_tdb_derivatives_against_40_digit_arithmetic():
    """Complex-step Jacobian and Richardson-extrapolated Hessian of the oracle against mpmath: the same RK4 map evaluated in
    40-digit arithmetic and differentiated numerically there (mp.diff), entry by entry."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(12)
    n, m, z, S = 2, 1, 5, 4
    G = rng.standard_normal((m + 1, n, n))
    Hc = 0.5 * rng.standard_normal((m + 1, n, n))
    it = O.TimeDependentBilinearIntegrator(0, n, n, m, 3, G, [("cos", 1.3, Hc)], 1, S).bind(z, 4)
    zz = rng.standard_normal(2 * z)
    zz[4] = 0.3
    mu = rng.standard_normal(n)
    Gm = [mp.matrix(G[j].tolist()) for j in range(m + 1)]
    Hm = [mp.matrix(Hc[j].tolist()) for j in range(m + 1)]

    def gen(u, t):
        return (Gm[0] + mp.cos(mp.mpf("1.3") * t) * Hm[0]) + u * (Gm[1] + mp.cos(mp.mpf("1.3") * t) * Hm[1])

    def flow(*v):
        x, uk, tk, dt, uk1 = mp.matrix(v[0:2]), v[2], v[3], v[4], v[z + 2]
        rhs = lambda tau, y: gen(uk + tau * (uk1 - uk), tk + tau * dt) * (y * dt)
        y, h = x, mp.mpf(1) / S
        for i in range(S):
            tau = i * h
            k1 = rhs(tau, y)
            k2 = rhs(tau + h / 2, y + h / 2 * k1)
            k3 = rhs(tau + h / 2, y + h / 2 * k2)
            k4 = rhs(tau + h, y + h * k3)
            y = y + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        return [v[z + i] - y[i] for i in range(n)]

    z0 = tuple(mp.mpf(float(v)) for v in zz)
    J, H = it.jac(zz), it.hess(zz, 0, mu)
    for a in range(2 * z):
        order = [0] * (2 * z)
        order[a] = 1
        for i in range(n):
            ref = mp.diff(lambda *v: flow(*v)[i], z0, tuple(order))
            assert abs(float(ref) - J[i, a]) <= 1e-13 * max(1.0, abs(float(ref))), (i, a)
    phi = lambda *v: sum(mp.mpf(float(mu[i])) * flow(*v)[i] for i in range(n))
    for a in range(2 * z):
        for b in range(a, 2 * z):
            order = [0] * (2 * z)
            order[a] += 1
            order[b] += 1
            ref = float(mp.diff(phi, z0, tuple(order)))
            assert abs(ref - H[a, b]) <= 2e-10 * max(1.0, abs(ref)), (a, b, ref, H[a, b])


def test_tdb_passes_the_reference_fd_bars():
    """test_integrator's finite-difference comparison (src/integrators/_integrators.jl:97-242) at the bar the reference's
    TimeDependentBilinearIntegrator tests use (atol 1e-3, time_dependent_bilinear_integrator.jl:256, :268)."""
    p = O.make_tdb_reference_carrier_problem()
    ev = O.OracleEvaluator(p)
    Z = p.Z0
    r, c = ev.jacobian_structure1()
    J = np.zeros((ev.n_constraints, p.n_vars))
    J[r - 1, c - 1] = ev.eval_constraint_jacobian(Z)
    eps = 1e-6
    Jfd = np.stack([(ev.eval_constraint(Z + eps * e) - ev.eval_constraint(Z - eps * e)) / (2 * eps) for e in np.eye(p.n_vars)], axis=1)
    assert np.allclose(J, Jfd, atol=1e-3) and np.abs(J - Jfd).max() <= 1e-8
