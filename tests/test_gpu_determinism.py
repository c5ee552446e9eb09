"""GPU: bit-reproducibility (the reference's evaluator is serial and deterministic, src/solvers/evaluator.jl:560-647).

Run to run, every callback returns the same bits without any option: fixed-order column sums (which drive the squaring counts
and the evaluation form of the matrix exponential), objective partial sums, listings that repeat a knot applied layer by
layer, global-variable entries accumulated in listing order.  Option "deterministic" = 1 additionally makes the bits independent
of `overlap_sweep` and of the entry-point family (host pointers / device pointers)."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, to_engine

pytestmark = pytest.mark.gpu


def _five(ev, Z, mu, sigma=0.7):
    out = {"f": np.array([ev.eval_objective(Z)])}
    g = np.empty(ev.shard.grad_len); ev.eval_objective_gradient(g, Z); out["grad"] = g
    c = np.empty(ev.shard.cons_len); ev.eval_constraint(c, Z); out["cons"] = c
    j = np.empty(ev.shard.jac_len); ev.eval_constraint_jacobian(j, Z); out["jac"] = j
    h = np.empty(ev.shard.hess_len); ev.eval_hessian_lagrangian(h, Z, sigma, mu); out["hess"] = h
    # J w and J' w (evaluator.jl:406-456): matrix-free on the general path, from the materialised slab on the small-state path and
    # with host-evaluated terms -- one writer per entry and fixed orders in both since round 4
    w = np.cos(np.arange(ev.n_variables) * 0.37)
    y = np.empty(ev.n_constraints); ev.eval_constraint_jacobian_product(y, Z, w); out["Jw"] = y
    yt = np.empty(ev.n_variables); ev.eval_constraint_jacobian_transpose_product(yt, Z, mu); out["JTw"] = yt
    return out


def _dup_constraint_problem():
    """A knot constraint whose `times` name knots twice (J' w: both listings add into the same entries of y, in listing order)."""
    p = O.make_scaled_problem(10, 40, 3, seed=23, with_constraint=True)
    c = p.constraints[0]
    p.constraints = [O.KnotConstraint(c.kind, c.comps, c.c, [2, 3, 3, 5, 2, 9], equality=c.equality)]
    return p


def _dup_problem():
    """Regularizer and knot objective that list knots twice (their contributions ADD, resp. the later listing overwrites), a
    DerivativeIntegrator next to the bilinear one: every accumulation site of the assembly sees several contributions."""
    p = O.make_scaled_problem(9, 40, 3, seed=17, with_constraint=True)
    n, m = 40, 3
    p.objectives = [O.QuadraticRegularizer(n, m, np.array([1.0, 0.7, 1.3]), times1=[1, 2, 3, 3, 5, 2, 9, 3]),
                    O.LinearRegularizer(n, m, np.array([0.3, -0.2, 0.9]), times1=[4, 4, 4, 7]),
                    O.KnotSqDistObjective(list(range(n)), [9, 9, 4], [1.0, 2.0, 0.5], np.random.default_rng(0).standard_normal((3, n)))]
    p.weights = [1.0, 0.4, 2.0]
    return p


@pytest.mark.parametrize("make", [lambda: O.make_scaled_problem(12, 40, 3, seed=21, with_constraint=True), _dup_problem, _dup_constraint_problem,
                                  lambda: O.make_global_problem(), lambda: O.make_scaled_problem(40, 130, 2, seed=2),
                                  lambda: O.make_standard_problem(N=10)],
                         ids=["general-path", "repeated-knots", "repeated-constraint-knots", "global-terms", "fused-sweep-128", "small-state-path"])
def test_repeated_calls_return_the_same_bits(make):
    import dto_amd
    p = make()
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p, "analytic"))
    try:
        rng = np.random.default_rng(3)
        Z = p.Z0 + 0.03 * rng.standard_normal(p.n_vars)
        mu = rng.standard_normal(ev.n_constraints)
        first = _five(ev, Z, mu)
        # the oracle agrees (so "the same bits" are the right ones) ...
        assert rel_err(first["jac"], ev_o.eval_constraint_jacobian(Z)) <= 1e-10
        assert rel_err(first["grad"], ev_o.eval_objective_gradient(Z)) <= 1e-10
        assert rel_err(first["hess"], ev_o.eval_hessian_lagrangian(Z, 0.7, mu)) <= 1e-8
        r1, c1 = ev_o.jacobian_structure1()
        import scipy.sparse as sp
        J = sp.csr_matrix((ev_o.eval_constraint_jacobian(Z), (r1 - 1, c1 - 1)), shape=(ev_o.n_constraints, p.n_vars))
        assert rel_err(first["Jw"], J @ np.cos(np.arange(p.n_vars) * 0.37)) <= 1e-10
        assert rel_err(first["JTw"], J.T @ mu) <= 1e-10
        # ... and every repetition reproduces them exactly, also with other points evaluated in between
        for rep in range(4):
            if rep == 2:
                _five(ev, Z + 0.1, mu)
            again = _five(ev, Z, mu)
            for k in first:
                assert np.array_equal(first[k], again[k]), (k, rep)
    finally:
        ev.close()


def test_deterministic_option_makes_the_bits_independent_of_how_they_are_asked_for():
    """256 states x 1700 knots: next to the chain the sweep would group its intervals by twelve instead of nine (option
    overlap_sweep), and the host-pointer Jacobian would run the chain in four chunks for its early hand-over."""
    import torch
    import dto_amd
    p = dto_amd.host.synthetic.make_scaled_problem(1700, 256, 4, seed=42)
    ev = dto_amd.Evaluator(p, eval_hessian=False)
    try:
        ev.set_option("deterministic", 1)
        Z = p.trajectory.vec()
        dev = torch.device("cuda", 0)
        dZ = torch.from_numpy(Z).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        outs = []
        for overlap in (1, 0, 1):
            ev.set_option("overlap_sweep", overlap)
            o = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
            ev.eval_jacobian_dev(dZ.data_ptr(), o.data_ptr(), st)
            torch.cuda.synchronize()
            outs.append(o)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        host = np.empty(ev.shard.jac_len)
        ev.eval_constraint_jacobian(host, Z)
        assert np.array_equal(host, outs[0].cpu().numpy())
    finally:
        ev.close()
