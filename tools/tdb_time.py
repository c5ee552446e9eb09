#!/usr/bin/env python3
"""Timing of the device TimeDependentBilinearIntegrator (a tools/ probe): Jacobian / Hessian per call, n states x N knots."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import dto_amd, dto_oracle as O
from helpers import to_engine
for n, N, sub in ((4, 1000, 16), (16, 500, 16), (32, 500, 16), (64, 200, 16)):
    po = O.make_tdb_problem(N=N, n=n, m=2, order=1, seed=3, substeps=sub)
    ev = dto_amd.Evaluator(to_engine(po))
    Z = po.Z0
    mu = np.ones(ev.n_constraints)
    j = np.empty(ev.shard.jac_len); h = np.empty(ev.shard.hess_len)
    out = []
    for name, fn in (("jac", lambda: ev.eval_constraint_jacobian(j, Z)), ("hess", lambda: ev.eval_hessian_lagrangian(h, Z, 1.0, mu))):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        out.append(f"{name} {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
    print(f"tdb {n} states x {N} knots, {sub} sub-steps (host pointers): " + ", ".join(out), flush=True)
    ev.close()
