"""GPU: the BASELINE.json configurations at their full sizes, checked through size-independent properties.

The oracle needs hours for 2000 x 256-state intervals, so each test samples a few intervals k of the full-size
engine output and compares the whole column block / diagonal Hessian block of knot k with the oracle run on the
TWO-KNOT sub-problem (z_k, z_{k+1}) with the same generators -- the bilinear and derivative integrators couple
nothing else, so block k of the big problem IS block 0 of that sub-problem.  On top: structural
zeros, identity blocks, and the chain-vs-sweep cross check E_k x_k = exp(A_k) x_k.

  configs[1]  64-state, 4 drives, N = 1000          Jacobian + constraints + Hessian
  configs[2]  256-state, N = 2000, Hessian enabled  Hessian (the Jacobian's test is in test_gpu_golden_and_shards.py)
  configs[3]  256-state, N = 16000 (one rank's view of the chunked chain: 3 chunks on one GPU), Jacobian + constraints
  configs[4]  1024-state + NonlinearInequality + L1 slack: f / grad / cons / Jacobian / Hessian on 3 knots vs the oracle,
              the (u,u) Hessian block vs central differences of the engine's own Jacobian at the reference's bar
              (evaluator.jl:790), and a 1024 x 64-knot run checked by sampled sub-problems
Tolerances as everywhere: 1e-10 max(1,|ref|) values / Jacobian, 1e-8 Hessian."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, to_engine

pytestmark = pytest.mark.gpu


def _sub_problem(G, Zk2, n, m, z, dt_idx, extra_objectives=()):
    """Oracle problem on the two knots (z_k, z_{k+1}) with the big problem's generators and objective terms."""
    return O.Problem(N=2, z=z, dt_idx=dt_idx,
                     integrators=[O.BilinearIntegrator(0, n, n, m, G), O.DerivativeIntegrator(n, m, n + m)],
                     objectives=[O.QuadraticRegularizer(n, m, np.ones(m))] + list(extra_objectives),
                     Z0=np.ascontiguousarray(Zk2).reshape(-1).copy())


def _dense(rows1, cols1, vals, shape):
    M = np.zeros(shape)
    M[rows1 - 1, cols1 - 1] = vals
    return M


def _jac_column_block(get, k, z, D, K):
    """Rows of column block k of the full-size Jacobian slab as a (cnt x z) array: per column and integrator the rows of
    interval k-1, then of interval k.  `get(lo, hi)` returns a host copy of vals[lo:hi]."""
    cnt = (1 if k >= 1 else 0) + (1 if k < K else 0)
    start = 0 if k == 0 else z * D + (k - 1) * 2 * z * D
    return get(start, start + z * cnt * D).reshape(z, cnt * D).T, cnt


def _check_jacobian_block(blk, cnt, k, sub_jac, n, m, z, K):
    """blk: (cnt*D x z) of knot k; sub_jac: dense (D x 2z) Jacobian of interval k from the two-knot oracle problem (rows:
    bilinear n, derivative m).  Own rows = its z_k half; previous interval's rows = the constant z_{k+1} half."""
    has_prev = k >= 1
    if k < K:
        own_b = blk[(n if has_prev else 0):(n if has_prev else 0) + n]
        own_d = blk[cnt * n + (m if has_prev else 0):cnt * n + (m if has_prev else 0) + m]
        assert rel_err(own_b, sub_jac[:n, :z]) <= 1e-10, ("bilinear rows", k, rel_err(own_b, sub_jac[:n, :z]))
        assert rel_err(own_d, sub_jac[n:, :z]) <= 1e-10, ("derivative rows", k)
        assert np.all(own_b[:, n + m:n + 2 * m] == 0.0)  # du columns: structural zeros that are still stored
    if has_prev:
        prev_b, prev_d = blk[:n], blk[cnt * n:cnt * n + m]
        ref_b = np.zeros((n, z)); ref_b[:, :n] = np.eye(n)
        ref_d = np.zeros((m, z)); ref_d[:, n:n + m] = np.eye(m)
        assert np.array_equal(prev_b, ref_b) and np.array_equal(prev_d, ref_d), ("z_{k+1} half", k)


def _hess_diag_block(get, k, z):
    """Upper triangle (incl. diagonal) of diagonal block k and the off-diagonal block (k-1, k) of the full-size Hessian."""
    tri = z * (z + 1) // 2
    if k == 0:
        return _tri_from_cols(get(0, tri), z), None
    start = tri + (k - 1) * (z * z + tri)
    v = get(start, start + z * z + tri)
    Hd, Ho = np.zeros((z, z)), np.zeros((z, z))
    pos = 0
    for b in range(z):
        Ho[:, b] = v[pos:pos + z]
        pos += z
        Hd[:b + 1, b] = v[pos:pos + b + 1]
        pos += b + 1
    return Hd, Ho


def _tri_from_cols(v, z):
    Hd = np.zeros((z, z))
    pos = 0
    for b in range(z):
        Hd[:b + 1, b] = v[pos:pos + b + 1]
        pos += b + 1
    return Hd


def _sampled_checks(prob_e, ev, n, m, ks, jac_get=None, hess_get=None, cons=None, mu=None, sigma=1.0, extra_objectives=()):
    """Compare column block / Hessian diagonal block of every sampled knot with the two-knot oracle problem."""
    traj = prob_e.trajectory
    N, z = traj.N, traj.dim
    K, D = N - 1, n + m
    dt_idx = traj.components[traj.timestep][0]
    G = prob_e.integrators[0].G
    X = traj.data
    for k in ks:
        if k < K:
            sub = _sub_problem(G, X[:, k:k + 2].T, n, m, z, dt_idx, extra_objectives)
            ev_o = O.OracleEvaluator(sub)
            r1, c1 = ev_o.jacobian_structure1()
            sub_jac = _dense(r1, c1, ev_o.eval_constraint_jacobian(sub.Z0), (D, 2 * z))
        if jac_get is not None:
            blk, cnt = _jac_column_block(jac_get, k, z, D, K)
            _check_jacobian_block(blk, cnt, k, sub_jac if k < K else None, n, m, z, K)
            if k < K and cons is not None:
                # chain vs sweep: the dense -E_k of the propagator chain applied to x_k equals exp(A_k) x_k of the sweep
                own = blk[(n if k >= 1 else 0):(n if k >= 1 else 0) + n]
                delta = cons[k * n:(k + 1) * n]
                assert rel_err(own[:, :n] @ X[:n, k], delta - X[:n, k + 1]) <= 1e-10, ("chain vs sweep", k)
        if cons is not None and k < K:
            ref = ev_o.eval_constraint(sub.Z0)
            assert rel_err(cons[k * n:(k + 1) * n], ref[:n]) <= 1e-10, ("bilinear defect", k)
            assert rel_err(cons[K * n + k * m:K * n + (k + 1) * m], ref[n:]) <= 1e-10, ("derivative defect", k)
        if hess_get is not None:
            Hd, Ho = _hess_diag_block(hess_get, k, z)
            if Ho is not None:
                assert np.all(Ho == 0.0), ("off-diagonal block", k)  # these integrators never fill it
            if k < K:
                mu_sub = np.concatenate([mu[k * n:(k + 1) * n], mu[K * n + k * m:K * n + (k + 1) * m]])
                r1, c1 = ev_o.hessian_structure1()
                Hs = _dense(r1, c1, ev_o.eval_hessian_lagrangian(sub.Z0, sigma, mu_sub), (2 * z, 2 * z))
                assert rel_err(Hd, Hs[:z, :z]) <= 1e-8, ("diagonal block", k, rel_err(Hd, Hs[:z, :z]))
                assert np.all(Hd[:n, :n] == 0.0)  # (x_k, x_k): identically zero for a defect linear in x
            else:
                # last knot: only the objective term of that knot
                sub = _sub_problem(G, X[:, k - 1:k + 1].T, n, m, z, dt_idx, extra_objectives)
                Hs = O.objective_full_hessian(sub, sub.Z0).toarray()[z:, z:]
                assert rel_err(Hd, sigma * np.triu(Hs)) <= 1e-8, ("last knot", k)


def _host_getter(vals):
    return lambda lo, hi: vals[lo:hi]


def test_config1_64_states_1000_knots():
    import dto_amd
    n, m, N = 64, 4, 1000
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m)
    ev = dto_amd.Evaluator(prob)
    Z = prob.trajectory.vec()
    mu = np.random.default_rng(1).standard_normal(ev.n_constraints)
    jac = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(jac, Z)
    cons = np.empty(ev.n_constraints); ev.eval_constraint(cons, Z)
    hes = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(hes, Z, 0.7, mu)
    assert np.isfinite(jac).all() and np.isfinite(cons).all() and np.isfinite(hes).all()
    _sampled_checks(prob, ev, n, m, (0, 1, 2, 311, 500, 998, 999), _host_getter(jac), _host_getter(hes), cons, mu, 0.7)
    ev.close()


def test_config2_hessian_256_states_2000_knots():
    """configs[2] names eval_hessian_lagrangian: sampled diagonal blocks against the oracle (second-order Frechet terms
    through the 3n x 3n block exponential), structural zeros of the off-diagonal blocks, accumulation of integrator,
    derivative and sigma * objective contributions on the same entries."""
    import dto_amd
    n, m, N = 256, 4, 2000
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m)
    ev = dto_amd.Evaluator(prob)
    Z = prob.trajectory.vec()
    mu = np.random.default_rng(2).standard_normal(ev.n_constraints)
    hes = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(hes, Z, 1.3, mu)
    assert np.isfinite(hes).all()
    _sampled_checks(prob, ev, n, m, (0, 1, 777, 1998, 1999), None, _host_getter(hes), None, mu, 1.3)
    # the whole vector: everything outside the diagonal blocks is zero
    z = n + 2 * m + 1
    tri = z * (z + 1) // 2
    body = hes[tri:].reshape(N - 1, z * z + tri)
    pos = 0
    off_idx = []
    for b in range(z):
        off_idx.append(np.arange(pos, pos + z))
        pos += z + b + 1
    assert np.all(body[:, np.concatenate(off_idx)] == 0.0)
    ev.close()


def test_config3_256_states_16000_knots_chunked_chain():
    """The 8-GPU configuration's total size on ONE device: 2.2e9 Jacobian values (17.6 GB) stay in HBM, the propagator
    chain runs in three workspace chunks; sampled column blocks (first / last of each chunk among them) come back for the
    comparison.  (The sharded form of this size is covered by the two-rank tests and the shard tests.)"""
    import torch
    import dto_amd
    n, m, N = 256, 4, 16000
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m)
    ev = dto_amd.Evaluator(prob, eval_hessian=False)
    dev = torch.device("cuda", 0)
    Z = prob.trajectory.vec()
    dZ = torch.from_numpy(Z).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    dj = torch.empty(ev.n_jacobian_entries, dtype=torch.float64, device=dev)
    dg = torch.empty(ev.n_constraints, dtype=torch.float64, device=dev)
    ev.eval_jacobian_dev(dZ.data_ptr(), dj.data_ptr(), st)
    ev.eval_constraint_dev(dZ.data_ptr(), dg.data_ptr(), st)
    torch.cuda.synchronize()
    assert ev.last_stats()[1] >= 1
    assert bool(torch.isfinite(dj).all()) and bool(torch.isfinite(dg).all())
    cons = dg.cpu().numpy()
    get = lambda lo, hi: dj[lo:hi].cpu().numpy()
    K = N - 1
    cap = int(36e9 / (9.0 * 256 * 256 * 8)) // 8 * 8  # the engine's chunk capacity at 256 states (dto_engine.cpp chunk_size)
    nchunk = -(-K // cap)
    per = min(cap, (-(-K // nchunk) + 7) // 8 * 8)
    assert nchunk == 3
    ks = (0, 1, per - 1, per, per + 1, 2 * per - 1, 2 * per, 9000, K - 1, K)
    _sampled_checks(prob, ev, n, m, ks, get, None, cons)
    del dj, dg
    ev.close()


def test_config4_workload_1024_states_three_knots_vs_oracle():
    """z = 1037 (x, u, du, s_du, dt), bilinear + derivative integrators, ||u|| - 1 <= 0 at 2:N-1,
    QuadraticRegularizer(:u) + LinearRegularizer(:s_du, 1e-2): every callback against the oracle on 3 knots."""
    import dto_amd
    n, m, N = 1024, 4, 3
    p = O.make_l1_slack_problem(N, n, m)
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p))
    assert ev.n_variables == p.n_vars == 3 * 1037 and ev.n_constraints == ev_o.n_constraints == 2 * (n + m) + 1
    r, c = ev.jacobian_structure(); r1, c1 = ev_o.jacobian_structure1()
    assert np.array_equal(r, r1) and np.array_equal(c, c1)
    r, c = ev.hessian_lagrangian_structure(); r1, c1 = ev_o.hessian_structure1()
    assert np.array_equal(r, r1) and np.array_equal(c, c1)
    lo, hi = ev.constraint_bounds(); lo_o, hi_o = ev_o.row_bounds()
    assert np.array_equal(lo, lo_o) and np.array_equal(hi, hi_o) and lo[-1] == -np.inf
    Z = p.Z0
    mu = np.random.default_rng(5).standard_normal(ev_o.n_constraints)
    assert rel_err(ev.eval_objective(Z), ev_o.eval_objective(Z)) <= 1e-10
    g = np.empty(ev.n_variables); ev.eval_objective_gradient(g, Z)
    assert rel_err(g, ev_o.eval_objective_gradient(Z)) <= 1e-10
    cons = np.empty(ev.n_constraints); ev.eval_constraint(cons, Z)
    assert rel_err(cons, ev_o.eval_constraint(Z)) <= 1e-10
    jac = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(jac, Z)
    e = rel_err(jac, ev_o.eval_constraint_jacobian(Z))
    assert e <= 1e-10, e
    hes = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(hes, Z, 0.9, mu)
    ref = ev_o.eval_hessian_lagrangian(Z, 0.9, mu, skip_uu=True)
    # (u_i, u_j) entries of the two interval knots: second-order Frechet terms, checked by differences below
    z = p.z
    uu = np.zeros(len(hes), dtype=bool)
    hr, hc = r1 - 1, c1 - 1
    for k in range(N - 1):
        inb = (hr >= k * z + n) & (hr < k * z + n + m) & (hc >= k * z + n) & (hc < k * z + n + m)
        uu |= inb
    assert uu.sum() == (N - 1) * m * (m + 1) // 2
    e = rel_err(hes[~uu], ref[~uu])
    assert e <= 1e-8, e
    # (u,u): mu' d2(delta)/du_i du_j + constraint + sigma * objective, by central differences of the engine's Jacobian and
    # gradient along u_i (the reference's own bar for Hessians is atol = 1e-2, evaluator.jl:790; differences of an
    # analytic Jacobian with h = 1e-4 give ~1e-6 here)
    jr, jc = ev.jacobian_structure()
    h = 1e-4
    for k in range(N - 1):
        for i in range(m):
            col = k * z + n + i
            Zp, Zm = Z.copy(), Z.copy()
            Zp[col] += h; Zm[col] -= h
            jp = np.empty_like(jac); ev.eval_constraint_jacobian(jp, Zp)
            jm = np.empty_like(jac); ev.eval_constraint_jacobian(jm, Zm)
            gp = np.empty_like(g); ev.eval_objective_gradient(gp, Zp)
            gm = np.empty_like(g); ev.eval_objective_gradient(gm, Zm)
            dJ = (jp - jm) / (2 * h)
            for j in range(i, m):
                colj = k * z + n + j
                sel = jc - 1 == colj
                fd = float(mu[jr[sel] - 1] @ dJ[sel]) + 0.9 * (gp[colj] - gm[colj]) / (2 * h)
                got = hes[(hr == col) & (hc == colj)]
                assert got.size == 1
                assert abs(got[0] - fd) <= 1e-5 * max(1.0, abs(fd)), (k, i, j, got[0], fd)
    ev.close()


def test_config4_workload_1024_states_64_knots_sampled():
    """The same workload on a longer horizon (the per-GPU share of N = 4000 over 8 ranks is 500 knots; 64 keep the host
    copies small): sampled knots against two-knot oracle problems, constraint rows and their Jacobian / Hessian entries
    against the closed forms of g(u) = ||u|| - 1."""
    import dto_amd
    n, m, N = 1024, 4, 64
    prob = dto_amd.host.synthetic.make_l1_slack_problem(N, n, m)
    ev = dto_amd.Evaluator(prob)
    Z = prob.trajectory.vec()
    z = prob.trajectory.dim
    assert z == 1037
    K, D = N - 1, n + m
    mu = np.random.default_rng(6).standard_normal(ev.n_constraints)
    cons = np.empty(ev.n_constraints); ev.eval_constraint(cons, Z)
    jac = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(jac, Z)
    assert np.isfinite(jac).all() and np.isfinite(cons).all()
    X = prob.trajectory.data
    U = X[n:n + m]
    # constraint rows (knots 2..N-1, 1-based) follow the dynamics rows
    nrm = np.linalg.norm(U[:, 1:N - 1], axis=0)
    assert rel_err(cons[K * D:], nrm - 1.0) <= 1e-10
    # Jacobian: the u columns of knots 2..N-1 carry one extra entry each (the constraint row, last in the column)
    colptr = np.zeros(N * z + 1, dtype=np.int64)
    cnt = np.full(N, 2); cnt[0] = 1; cnt[-1] = 1
    per = np.repeat(cnt * D, z).astype(np.int64)
    extra = np.zeros(N * z, dtype=np.int64)
    for kn in range(1, N - 1):
        extra[kn * z + n:kn * z + n + m] = 1
    colptr[1:] = np.cumsum(per + extra)
    assert colptr[-1] == ev.n_jacobian_entries
    for kn in (1, 2, 33, N - 2):
        for j in range(m):
            c = kn * z + n + j
            assert rel_err(jac[colptr[c + 1] - 1], U[j, kn] / nrm[kn - 1]) <= 1e-10
    # integrator blocks at sampled knots (own rows via the two-knot oracle problem; the column blocks are cut by colptr)
    G = prob.integrators[0].G
    lin = [O.LinearRegularizer(n + 2 * m, m, np.full(m, 1e-2))]
    for k in (0, 1, 40, K - 1):
        sub = _sub_problem(G, X[:, k:k + 2].T, n, m, z, n + 3 * m, lin)
        ev_o = O.OracleEvaluator(sub)
        r1, c1 = ev_o.jacobian_structure1()
        sub_jac = _dense(r1, c1, ev_o.eval_constraint_jacobian(sub.Z0), (D, 2 * z))
        has_prev = k >= 1
        for j in list(range(0, n, 97)) + list(range(n, z)):
            c = k * z + j
            col = jac[colptr[c]:colptr[c + 1]]
            own_b = col[(n if has_prev else 0):(n if has_prev else 0) + n]
            assert rel_err(own_b, sub_jac[:n, j]) <= 1e-10, (k, j)
            ncnt = 2 if has_prev else 1
            own_d = col[ncnt * n + (m if has_prev else 0):ncnt * n + (m if has_prev else 0) + m]
            assert rel_err(own_d, sub_jac[n:, j]) <= 1e-10, (k, j)
        ref = ev_o.eval_constraint(sub.Z0)
        assert rel_err(cons[k * n:(k + 1) * n], ref[:n]) <= 1e-10
    ev.close()


def test_config4_per_rank_share_1024_states_500_knots_of_4000():
    """BASELINE configs[4] as ONE of its eight ranks runs it: the 4000-knot problem (1024 states, ||u|| - 1 <= 0, L1 slack), the
    handle owning knots 1501..2000 -- 500 knots, the per-GPU share -- with every output resident in HBM (8.5 GB of Jacobian
    values, 6.4 GB of Hessian values).  Sampled knots of the shard against two-knot oracle problems: bilinear and derivative
    defects, Jacobian column blocks cut by the GLOBAL column pointers minus the shard's offset, the constraint rows and their
    Jacobian entries against the closed form, one Hessian diagonal block ((u, u) entries left to the three-knot test above)."""
    import torch
    import dto_amd
    n, m, N, world, rank = 1024, 4, 4000, 8, 3
    prob = dto_amd.host.synthetic.make_l1_slack_problem(N, n, m)
    lo, hi = dto_amd.distributed.shard_ranges(N, world)[rank]
    assert (lo, hi) == (1501, 2000)
    ev = dto_amd.Evaluator(prob, k_lo=lo, k_hi=hi)
    sh = ev.shard
    z = prob.trajectory.dim
    K, D = N - 1, n + m
    dev = torch.device("cuda", 0)
    Z = prob.trajectory.vec()
    dZ = torch.from_numpy(Z).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    mu = np.random.default_rng(8).standard_normal(ev.n_constraints)
    dmu = torch.from_numpy(mu).to(dev)
    dj = torch.empty(sh.jac_len, dtype=torch.float64, device=dev)
    dg = torch.empty(sh.cons_len, dtype=torch.float64, device=dev)
    ev.eval_jacobian_dev(dZ.data_ptr(), dj.data_ptr(), st)
    ev.eval_constraint_dev(dZ.data_ptr(), dg.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dj).all()) and bool(torch.isfinite(dg).all())
    cons = dg.cpu().numpy()
    X = prob.trajectory.data
    U = X[n:n + m]
    n_knots = hi - lo + 1
    n_int = n_knots                                    # every owned knot of an interior rank owns an interval
    assert sh.cons_len == n_int * D + n_knots          # dynamics rows of the owned intervals + one norm row per owned knot
    k0 = lo - 1                                        # first owned knot, 0-based
    nrm = np.linalg.norm(U[:, k0:k0 + n_knots], axis=0)
    assert rel_err(cons[n_int * D:], nrm - 1.0) <= 1e-10
    # global column pointers of the reference's CSC order: per column cnt * D integrator rows + the constraint row of u columns
    cnt = np.full(N, 2); cnt[0] = 1; cnt[-1] = 1
    per = np.repeat(cnt * D, z).astype(np.int64)
    extra = np.zeros(N * z, dtype=np.int64)
    for kn in range(1, N - 1):
        extra[kn * z + n:kn * z + n + m] = 1
    colptr = np.zeros(N * z + 1, dtype=np.int64)
    colptr[1:] = np.cumsum(per + extra)
    assert colptr[-1] == ev.n_jacobian_entries and colptr[k0 * z] == sh.jac_lo and colptr[(k0 + n_knots) * z] == sh.jac_lo + sh.jac_len
    col = lambda c: dj[colptr[c] - sh.jac_lo:colptr[c + 1] - sh.jac_lo].cpu().numpy()
    for kn in (k0, k0 + 1, k0 + 250, k0 + n_knots - 1):
        for j in range(m):
            assert rel_err(col(kn * z + n + j)[-1], U[j, kn] / nrm[kn - k0]) <= 1e-10   # d(||u|| - 1)/du_j, last in its column
    G = prob.integrators[0].G
    lin = [O.LinearRegularizer(n + 2 * m, m, np.full(m, 1e-2))]
    for k in (k0, k0 + 251, k0 + n_knots - 1):
        sub = _sub_problem(G, X[:, k:k + 2].T, n, m, z, n + 3 * m, lin)
        ev_o = O.OracleEvaluator(sub)
        r1, c1 = ev_o.jacobian_structure1()
        sub_jac = _dense(r1, c1, ev_o.eval_constraint_jacobian(sub.Z0), (D, 2 * z))
        for j in list(range(0, n, 131)) + list(range(n, z)):
            cj = col(k * z + j)
            own_b = cj[n:2 * n]                        # rows of interval k-1 come first (every sampled knot has one)
            assert rel_err(own_b, sub_jac[:n, j]) <= 1e-10, (k, j)
            own_d = cj[2 * n + m:2 * n + 2 * m]
            assert rel_err(own_d, sub_jac[n:, j]) <= 1e-10, (k, j)
        ref = ev_o.eval_constraint(sub.Z0)
        kl = k - k0
        assert rel_err(cons[kl * n:(kl + 1) * n], ref[:n]) <= 1e-10
        assert rel_err(cons[n_int * n + kl * m:n_int * n + (kl + 1) * m], ref[n:]) <= 1e-10
    del dj
    # Hessian of the shard: one diagonal block against the oracle (mu of the interval's rows; the norm constraint of knot k adds
    # its own (u, u) block, which lies among the entries left out)
    dh = torch.empty(sh.hess_len, dtype=torch.float64, device=dev)
    ev.eval_hessian_dev(dZ.data_ptr(), 0.8, dmu.data_ptr(), dh.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dh).all())
    tri = z * (z + 1) // 2
    k = k0 + 123
    start = tri + (k - 1) * (z * z + tri) - sh.hess_lo     # column block of knot k: per column b the z off-diagonal rows, then b + 1
    blkv = dh[start:start + z * z + tri].cpu().numpy()
    Hd = np.zeros((z, z))
    pos = 0
    for b in range(z):
        pos += z
        Hd[:b + 1, b] = blkv[pos:pos + b + 1]
        pos += b + 1
    sub = _sub_problem(G, X[:, k:k + 2].T, n, m, z, n + 3 * m, lin)
    ev_o = O.OracleEvaluator(sub)
    mu_sub = np.concatenate([mu[k * n:(k + 1) * n], mu[K * n + k * m:K * n + (k + 1) * m]])
    r1, c1 = ev_o.hessian_structure1()
    Hs = _dense(r1, c1, ev_o.eval_hessian_lagrangian(sub.Z0, 0.8, mu_sub, skip_uu=True), (2 * z, 2 * z))[:z, :z]
    keep = np.ones((z, z), dtype=bool)
    keep[n:n + m, n:n + m] = False
    assert rel_err(Hd[keep], Hs[keep]) <= 1e-8, rel_err(Hd[keep], Hs[keep])
    assert np.all(Hd[:n, :n] == 0.0)
    ev.close()
