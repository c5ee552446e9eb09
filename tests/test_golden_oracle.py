"""The oracle reproduces the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest

import dto_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.CASES


CASES = load_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    prob = CASES[name]()
    ev = O.OracleEvaluator(prob)
    jr, jc = ev.jacobian_structure1()
    hr, hc = ev.hessian_structure1()
    assert np.array_equal(jr, g["jac_rows"]) and np.array_equal(jc, g["jac_cols"])
    assert np.array_equal(hr, g["hess_rows"]) and np.array_equal(hc, g["hess_cols"])
    Z, mu, sigma = g["Z"], g["mu"], float(g["sigma"])
    assert ev.eval_objective(Z) == pytest.approx(float(g["f"]), rel=1e-13, abs=1e-15)
    for key, val in (("grad", ev.eval_objective_gradient(Z)), ("cons", ev.eval_constraint(Z)),
                     ("jac", ev.eval_constraint_jacobian(Z)), ("hess", ev.eval_hessian_lagrangian(Z, sigma, mu))):
        assert np.allclose(val, g[key], rtol=1e-12, atol=1e-13), key
