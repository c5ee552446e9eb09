"""Randomised problem layouts (fixed seeds): component blocks in shuffled order inside the knot, state sizes across
the three execution paths (one wavefront, four wavefronts, batched GEMM), random time lists with repeats, weights,
baselines, both constraint kinds -- every callback and the index structure against the oracle.  Index arithmetic
that only works for the canonical x,u,du,dt order would show up here."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, run_all, to_engine

pytestmark = pytest.mark.gpu


def random_problem(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 5, 9, 16, 17, 24, 33, 40]))
    m = int(rng.integers(1, 4))
    N = int(rng.integers(3, 8))
    fill = int(rng.integers(0, 3))
    blocks = [("x", n), ("u", m), ("du", m), ("dt", 1)] + ([("f", fill)] if fill else [])
    order = rng.permutation(len(blocks))
    off, pos = {}, 0
    for i in order:
        name, d = blocks[i]
        off[name] = pos
        pos += d
    z = pos
    G = rng.standard_normal((m + 1, n, n)) / np.sqrt(n) * float(rng.choice([0.3, 1.0, 3.0]))
    data = np.zeros((z, N))
    data[off["x"]:off["x"] + n] = rng.standard_normal((n, N))
    data[off["u"]:off["u"] + m] = 0.3 * rng.standard_normal((m, N))
    data[off["du"]:off["du"] + m] = rng.standard_normal((m, N))
    data[off["dt"]] = 0.05 + 0.2 * rng.random(N)
    if fill:
        data[off["f"]:off["f"] + fill] = rng.standard_normal((fill, N))
    integrators = [O.BilinearIntegrator(off["x"], n, off["u"], m, G), O.DerivativeIntegrator(off["u"], m, off["du"])]
    if rng.random() < 0.5:
        integrators.reverse()

    def times(k):
        return [int(t) for t in rng.integers(1, N + 1, size=k)]

    objectives = [O.QuadraticRegularizer(off["u"], m, 0.5 + rng.random(m),
                                         baseline=rng.standard_normal((m, N)) if rng.random() < 0.5 else None,
                                         times1=sorted(set(times(N))) if rng.random() < 0.5 else None),
                  O.LinearRegularizer(off["du"], m, rng.standard_normal(m)),
                  O.MinimumTimeObjective(float(1.0 + rng.random())),
                  O.KnotSqDistObjective(list(range(off["x"], off["x"] + n)), times(3), list(0.5 + rng.random(3)),
                                        rng.standard_normal((3, n)))]
    weights = list(0.3 + rng.random(len(objectives)))
    constraints = [O.KnotConstraint("norm", list(range(off["u"], off["u"] + m)), 1.0, times(4), equality=False),
                   O.KnotConstraint("sqnorm", list(range(off["du"], off["du"] + m)), 0.7, times(2), equality=True)]
    return O.Problem(N=N, z=z, dt_idx=off["dt"], integrators=integrators, objectives=objectives, weights=weights,
                     constraints=constraints, Z0=data.T.reshape(-1).copy())


@pytest.mark.parametrize("seed", list(range(100, 112)))
def test_random_layout(seed):
    import dto_amd
    p = random_problem(seed)
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p))
    try:
        jr, jc = ev.jacobian_structure()
        r1, c1 = ev_o.jacobian_structure1()
        assert np.array_equal(jr, r1) and np.array_equal(jc, c1)
        hr, hc = ev.hessian_lagrangian_structure()
        r1, c1 = ev_o.hessian_structure1()
        assert np.array_equal(hr, r1) and np.array_equal(hc, c1)
        rng = np.random.default_rng(seed + 1)
        Z = p.Z0 + 0.02 * rng.standard_normal(p.n_vars)
        mu = rng.standard_normal(ev_o.n_constraints)
        out = run_all(ev, p, Z, mu, sigma=0.6)
        errs = {"f": rel_err(out["f"], ev_o.eval_objective(Z)), "grad": rel_err(out["grad"], ev_o.eval_objective_gradient(Z)),
                "cons": rel_err(out["cons"], ev_o.eval_constraint(Z)), "jac": rel_err(out["jac"], ev_o.eval_constraint_jacobian(Z)),
                "hess": rel_err(out["hess"], ev_o.eval_hessian_lagrangian(Z, 0.6, mu))}
        for k, v in errs.items():
            assert v <= (1e-8 if k == "hess" else 1e-10), (seed, k, v, p.N, p.z)
    finally:
        ev.close()
