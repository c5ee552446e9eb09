"""Drop-in check at the level the reference is used: an interior-point/SQP-type solver drives the MOI callbacks
(BASELINE configs[0]: README 2-state bilinear problem, N = 50, QuadraticRegularizer).  Ipopt is not available here,
so SciPy's `trust-constr` stands in for it; it consumes exactly what Ipopt/MadNLP consume -- objective, gradient,
constraint values, sparse Jacobian in the evaluator's structure, Hessian of the Lagrangian -- once from the engine
and once from the oracle, and must walk the same iterates."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import Bounds, NonlinearConstraint, minimize

import dto_oracle as O
from helpers import to_engine

pytestmark = pytest.mark.gpu


class _EngineCallbacks:
    def __init__(self, ev):
        self.ev = ev
        self.jr, self.jc = ev.jacobian_structure()
        self.hr, self.hc = ev.hessian_lagrangian_structure()
        self.n, self.m = ev.n_variables, ev.n_constraints

    def f(self, Z):
        return self.ev.eval_objective(Z)

    def grad(self, Z):
        g = np.empty(self.n)
        self.ev.eval_objective_gradient(g, Z)
        return g

    def cons(self, Z):
        c = np.empty(self.m)
        self.ev.eval_constraint(c, Z)
        return c

    def jac(self, Z):
        v = np.empty(self.jr.size)
        self.ev.eval_constraint_jacobian(v, Z)
        return sp.csr_matrix((v, (self.jr - 1, self.jc - 1)), shape=(self.m, self.n))

    def hess(self, Z, sigma, mu):
        v = np.empty(self.hr.size)
        self.ev.eval_hessian_lagrangian(v, Z, sigma, mu)
        U = sp.csr_matrix((v, (self.hr - 1, self.hc - 1)), shape=(self.n, self.n))
        return U + sp.triu(U, 1).T


class _OracleCallbacks(_EngineCallbacks):
    def __init__(self, ev):
        self.ev = ev
        self.jr, self.jc = ev.jacobian_structure1()
        self.hr, self.hc = ev.hessian_structure1()
        self.n, self.m = ev.prob.n_vars, ev.n_constraints

    def f(self, Z):
        return self.ev.eval_objective(Z)

    def grad(self, Z):
        return self.ev.eval_objective_gradient(Z)

    def cons(self, Z):
        return self.ev.eval_constraint(Z)

    def jac(self, Z):
        return sp.csr_matrix((self.ev.eval_constraint_jacobian(Z), (self.jr - 1, self.jc - 1)), shape=(self.m, self.n))

    def hess(self, Z, sigma, mu):
        v = self.ev.eval_hessian_lagrangian(Z, sigma, mu)
        U = sp.csr_matrix((v, (self.hr - 1, self.hc - 1)), shape=(self.n, self.n))
        return U + sp.triu(U, 1).T


def _solve(cb, Z0, lo, hi, p):
    iterates = []
    con = NonlinearConstraint(cb.cons, lo, hi, jac=cb.jac, hess=lambda Z, v: cb.hess(Z, 0.0, v))
    # the reference's problems bound dt and the controls through MOI variable bounds: keep dt positive here
    lb = np.full(cb.n, -np.inf)
    ub = np.full(cb.n, np.inf)
    lb[p.dt_idx:p.z * p.N:p.z] = 0.05
    ub[p.dt_idx:p.z * p.N:p.z] = 0.2
    res = minimize(cb.f, Z0, jac=cb.grad, hess=lambda Z: cb.hess(Z, 1.0, np.zeros(cb.m)), constraints=[con],
                   bounds=Bounds(lb, ub), method="trust-constr",
                   options={"maxiter": 25, "gtol": 1e-10, "xtol": 1e-12, "sparse_jacobian": True},
                   callback=lambda Z, state: iterates.append(np.array(Z)) and False)
    return res, iterates


def test_solver_drives_engine_and_oracle_to_the_same_iterates():
    import dto_amd
    p = O.make_readme_problem()
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p))
    try:
        lo, hi = ev.constraint_bounds()
        res_e, it_e = _solve(_EngineCallbacks(ev), p.Z0.copy(), lo, hi, p)
        res_o, it_o = _solve(_OracleCallbacks(ev_o), p.Z0.copy(), lo, hi, p)
        assert len(it_e) == len(it_o) and len(it_e) >= 5
        for a, b in zip(it_e, it_o):
            assert np.allclose(a, b, rtol=1e-7, atol=1e-7)
        assert abs(res_e.fun - res_o.fun) <= 1e-8 * max(1.0, abs(res_o.fun))
        # the solver made progress on the defects (both paths)
        c0 = np.empty(ev.n_constraints); ev.eval_constraint(c0, p.Z0)
        c1 = np.empty(ev.n_constraints); ev.eval_constraint(c1, res_e.x)
        assert np.abs(c1).max() < 0.5 * np.abs(c0).max()
    finally:
        ev.close()
