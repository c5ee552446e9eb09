"""GPU: the HIP engine against the committed golden vectors, sharded handles, and size-independent
properties at the BASELINE sizes."""
import os

import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, to_engine

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.CASES


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_matches_golden(name):
    import dto_amd
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ev = dto_amd.Evaluator(to_engine(CASES[name](), "analytic"))
    try:
        r, c = ev.jacobian_structure()
        assert np.array_equal(r, g["jac_rows"]) and np.array_equal(c, g["jac_cols"])
        r, c = ev.hessian_lagrangian_structure()
        assert np.array_equal(r, g["hess_rows"]) and np.array_equal(c, g["hess_cols"])
        Z, mu, sigma = g["Z"], g["mu"], float(g["sigma"])
        assert rel_err(ev.eval_objective(Z), float(g["f"])) <= 1e-10
        out = np.empty(ev.n_variables); ev.eval_objective_gradient(out, Z); assert rel_err(out, g["grad"]) <= 1e-10
        out = np.empty(ev.n_constraints); ev.eval_constraint(out, Z); assert rel_err(out, g["cons"]) <= 1e-10
        out = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(out, Z); assert rel_err(out, g["jac"]) <= 1e-10
        out = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(out, Z, sigma, mu); assert rel_err(out, g["hess"]) <= 1e-8
    finally:
        ev.close()


def test_sharded_handles_reproduce_the_full_vectors():
    import dto_amd
    p = O.make_scaled_problem(11, 8, 2, seed=3, with_constraint=True)
    ev_o = O.OracleEvaluator(p)
    Z = p.Z0
    mu = np.random.default_rng(2).standard_normal(ev_o.n_constraints)
    ref = {"jac": ev_o.eval_constraint_jacobian(Z), "hess": ev_o.eval_hessian_lagrangian(Z, 0.3, mu),
           "grad": ev_o.eval_objective_gradient(Z), "cons": ev_o.eval_constraint(Z)}
    for world in (2, 3):
        got = {k: np.full_like(v, np.nan) for k, v in ref.items()}
        f = 0.0
        for lo, hi in dto_amd.distributed.shard_ranges(p.N, world):
            ev = dto_amd.Evaluator(to_engine(p), k_lo=lo, k_hi=hi)
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
            assert rel_err(got[k], ref[k]) <= (1e-8 if k == "hess" else 1e-10), (world, k)


def test_device_pointer_entry_points_match_host_entry_points():
    import torch
    import dto_amd
    p = O.make_scaled_problem(6, 16, 3, seed=12, with_constraint=True)
    ev = dto_amd.Evaluator(to_engine(p))
    Z = p.Z0
    mu = np.random.default_rng(3).standard_normal(ev.n_constraints)
    dev = torch.device("cuda", 0)
    dZ, dmu = torch.from_numpy(Z).to(dev), torch.from_numpy(mu).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    dj = torch.empty(ev.n_jacobian_entries, dtype=torch.float64, device=dev)
    dh = torch.empty(ev.n_hessian_entries, dtype=torch.float64, device=dev)
    dg = torch.empty(ev.n_constraints, dtype=torch.float64, device=dev)
    dgr = torch.empty(ev.n_variables, dtype=torch.float64, device=dev)
    df = torch.empty(1, dtype=torch.float64, device=dev)
    ev.eval_jacobian_dev(dZ.data_ptr(), dj.data_ptr(), st)
    ev.eval_hessian_dev(dZ.data_ptr(), 0.9, dmu.data_ptr(), dh.data_ptr(), st)
    ev.eval_constraint_dev(dZ.data_ptr(), dg.data_ptr(), st)
    ev.eval_gradient_dev(dZ.data_ptr(), dgr.data_ptr(), st)
    ev.eval_objective_dev(dZ.data_ptr(), df.data_ptr(), st)
    torch.cuda.synchronize()
    j = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(j, Z)
    h = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(h, Z, 0.9, mu)
    g = np.empty(ev.n_constraints); ev.eval_constraint(g, Z)
    gr = np.empty(ev.n_variables); ev.eval_objective_gradient(gr, Z)
    assert np.array_equal(dj.cpu().numpy(), j) and np.array_equal(dg.cpu().numpy(), g)
    assert np.array_equal(dgr.cpu().numpy(), gr) and df.item() == ev.eval_objective(Z)
    assert rel_err(dh.cpu().numpy(), h) <= 1e-13  # Hessian accumulates with float atomics
    ev.close()


def test_jacobian_into_an_8_byte_aligned_device_buffer():
    """The C ABI promises nothing about the alignment of the caller's value buffer beyond that of a double: the general
    path (40 states; its zero-fill uses 16-byte stores where it can) must write the same numbers at an odd offset."""
    import torch
    import dto_amd
    p = O.make_scaled_problem(5, 40, 2, seed=4, with_constraint=True)
    ev = dto_amd.Evaluator(to_engine(p), eval_hessian=False)
    dev = torch.device("cuda", 0)
    dZ = torch.from_numpy(p.Z0).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = ev.n_jacobian_entries
    a = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((n + 3,), float("nan"), dtype=torch.float64, device=dev)
    ev.eval_jacobian_dev(dZ.data_ptr(), a.data_ptr(), st)
    ev.eval_jacobian_dev(dZ.data_ptr(), b.data_ptr() + 8, st)
    torch.cuda.synchronize()
    assert a.data_ptr() % 16 == 0
    assert torch.equal(a, b[1:n + 1]) and bool(torch.isfinite(a).all())
    assert bool(torch.isnan(b[0])) and bool(torch.isnan(b[n + 1:]).all())  # nothing written outside the slab
    ev.close()


def test_full_size_properties_256x2000():
    """BASELINE configs[2] size: parity through size-independent properties (the oracle would need
    hours here): (1) the x_k Jacobian block of interval k is -exp(dt G(u_k)), checked against the
    oracle on a few sampled intervals; (2) linearity of the defect in x: delta(x) Jacobian-vector
    consistency  J_x x_k + x_{k+1} = delta_k ... i.e. E_k x_k from the matrix chain equals exp(A)x from
    the independent generator sweep; (3) structural zeros are zero, identity blocks are identity."""
    import scipy.linalg as sla
    import dto_amd
    n, m, N = 256, 4, 2000
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m)
    ev = dto_amd.Evaluator(prob, eval_hessian=False)
    Z = prob.trajectory.vec()
    z, K, D = n + 2 * m + 1, N - 1, n + m
    vals = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(vals, Z)
    cons = np.empty(ev.n_constraints); ev.eval_constraint(cons, Z)
    assert np.isfinite(vals).all() and np.isfinite(cons).all()
    G = prob.integrators[0].G
    X = prob.trajectory.data
    worst_chain = worst_sweep = 0.0
    for k in (0, 1, 777, 1500, K - 1):  # 0-based interval
        colstart = 0 if k == 0 else z * D + (k - 1) * 2 * z * D
        cnt = D if k == 0 else 2 * D
        blk = vals[colstart:colstart + z * cnt].reshape(z, cnt).T  # rows of the column block x knot comps
        own = blk[(0 if k == 0 else n):(0 if k == 0 else n) + n, :] if k == 0 else blk[n:2 * n, :]
        u, dt, x = X[n:n + m, k], X[z - 1, k], X[:n, k]
        A = dt * (G[0] + np.tensordot(u, G[1:], axes=(0, 0)))
        E = sla.expm(A)
        assert rel_err(own[:, :n], -E) <= 1e-10
        for j in range(m):
            Lx = sla.expm_frechet(A, dt * G[1 + j], compute_expm=False) @ x
            assert rel_err(own[:, n + j], -Lx) <= 1e-10
        assert rel_err(own[:, z - 1], -(A / dt) @ (E @ x)) <= 1e-10
        assert np.all(own[:, n + m:n + 2 * m] == 0.0)  # du columns: structural zeros, still stored
        if k >= 1:
            prev = blk[:n, :]
            assert np.array_equal(prev[:, :n], np.eye(n)) and np.all(prev[:, n:] == 0.0)
        delta = cons[k * n:(k + 1) * n]
        worst_sweep = max(worst_sweep, rel_err(delta, X[:n, k + 1] - E @ x))
        worst_chain = max(worst_chain, rel_err(own[:, :n] @ x, delta - X[:n, k + 1]))
    assert worst_sweep <= 1e-10 and worst_chain <= 1e-10
    ev.close()
