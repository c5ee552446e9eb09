#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the oracle (oracle/dto_oracle.py).

The reference's own tests hold no literal expected values for this path, and the Julia
reference cannot run here, so these fixtures freeze the ORACLE's answers (inputs + expected outputs,
fp64): they guard the oracle against drift and give the GPU tests a file-based second anchor.
Run:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import dto_oracle as O  # noqa: E402

CASES = {
    "readme_n2_N50": lambda: O.make_readme_problem(),
    "standard_n4_N10": lambda: O.make_standard_problem(N=10),
    "type1_derivative": lambda: O.make_type1_derivative_problem(),
    "scaled_n8_m2_N6": lambda: O.make_scaled_problem(6, 8, 2, seed=14, with_constraint=True),
    "scaled_n16_m4_N5": lambda: O.make_scaled_problem(5, 16, 4, seed=21, with_constraint=True),
    "skew_n32_m2_N4": lambda: O.make_scaled_problem(4, 32, 2, seed=5, skew=True),
    # host-evaluated / merged term kinds and the built-in coherent-fidelity loss
    "ket_infidelity_N8": lambda: O.make_ket_problem(),
    "closure_terms_N9": lambda: O.make_closure_problem(),
    "external_integrator_N7": lambda: O.make_external_integrator_problem(),
    "global_terms_N7": lambda: O.make_global_problem(),
    # TimeDependentBilinearIntegrator (oracle: RK4 restatement of the reference's ODE, pinned against DOP853 at rtol 1e-12 in
    # tests/test_oracle_pinning.py): controls interpolated / held, the reference's own carrier test, a 16-state case
    "tdb_modulated_N6": lambda: O.make_tdb_problem(N=6, n=4, m=2, order=1, seed=5, substeps=16),
    "tdb_modulated_order0_N6": lambda: O.make_tdb_problem(N=6, n=4, m=2, order=0, seed=6, substeps=16, with_derivative=True),
    "tdb_reference_carrier_N10": lambda: O.make_tdb_reference_carrier_problem(),
    "tdb_n16_N3": lambda: O.make_tdb_problem(N=3, n=16, m=2, order=1, seed=7, substeps=8, n_mods=1),
}


def expected(prob, seed=0):
    ev = O.OracleEvaluator(prob)
    rng = np.random.default_rng(seed)
    Z = prob.Z0 + 0.05 * rng.standard_normal(prob.n_vars)
    mu = rng.standard_normal(ev.n_constraints)
    sigma = 0.7
    jr, jc = ev.jacobian_structure1()
    hr, hc = ev.hessian_structure1()
    lo, hi = ev.row_bounds()
    return dict(Z=Z, mu=mu, sigma=np.array(sigma), f=np.array(ev.eval_objective(Z)),
                grad=ev.eval_objective_gradient(Z), cons=ev.eval_constraint(Z),
                jac_rows=jr, jac_cols=jc, jac=ev.eval_constraint_jacobian(Z),
                hess_rows=hr, hess_cols=hc, hess=ev.eval_hessian_lagrangian(Z, sigma, mu),
                row_lo=lo, row_hi=hi)


if __name__ == "__main__":
    for name, make in CASES.items():
        out = expected(make())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items() if k in ("Z", "jac", "hess")})
