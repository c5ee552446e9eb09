"""Random 33..64-state problems against the oracle: time steps over two decades (both evaluation forms, 0..many squarings, sub-stepped
sweeps), 1..5 generators, 2..40 knots, skew and general generators; prints the worst relative error per callback.
usage (through gpurun): python tools/fuzz_states64.py [cases=60]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np
import dto_amd
import dto_oracle as O
from helpers import to_engine, run_all


def relmax(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(64)
worst = {"cons": 0, "jac": 0, "hess": 0}
for it in range(cases):
    n = int(rng.integers(33, 65)); m = int(rng.integers(1, 5)); N = int(rng.choice([2, 3, 5, 9, 17, 40]))
    skew = bool(rng.integers(0, 2))
    p = O.make_scaled_problem(N, n, m, seed=int(rng.integers(1, 10**6)), skew=skew, with_constraint=bool(rng.integers(0, 2)))
    Z = p.Z0.copy()
    base = 0.1 * np.sqrt(256.0 / n) * (1.0 if not skew else 4.0)
    Z[p.dt_idx::p.z] = base * 10 ** rng.uniform(-1.5, 0.9, p.N)
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p))
    mu = rng.standard_normal(ev_o.n_constraints)
    out = run_all(ev, p, Z, mu, sigma=0.8)
    e = {"cons": relmax(out["cons"], ev_o.eval_constraint(Z)), "jac": relmax(out["jac"], ev_o.eval_constraint_jacobian(Z)),
         "hess": relmax(out["hess"], ev_o.eval_hessian_lagrangian(Z, 0.8, mu))}
    st = ev.last_stats(); ev.close()
    for k in e:
        worst[k] = max(worst[k], e[k])
    print(f"{it:2d} n={n} m={m} N={N} skew={skew} stats={st} " + " ".join(f"{k} {v:.1e}" for k, v in e.items()), flush=True)
print("WORST", worst)
assert worst["cons"] <= 1e-9 and worst["jac"] <= 1e-9 and worst["hess"] <= 1e-7, worst
