"""How much accuracy does the sub-stepping threshold of the generator sweep buy?  Runs the 1024-state problem (growth rate
alpha_3 ~ 10) with the sweep in one round (DTO_THETA_V=10.5) and in two (default 9) in child processes and compares both
with the oracle on the first intervals."""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/oracle"); sys.path.insert(0, %r + "/tests")
import dto_oracle as O, dto_amd
from helpers import to_engine
p = O.make_scaled_problem(3, 1024, 4, seed=42)
ev = dto_amd.Evaluator(to_engine(p), eval_hessian=True)
Z = p.Z0
c = np.empty(ev.n_constraints); ev.eval_constraint(c, Z)
j = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(j, Z)
mu = np.ones(ev.n_constraints); h = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(h, Z, 1.0, mu)
np.savez(sys.argv[1], c=c, j=j, h=h); print("stats", ev.last_stats())
''' % (ROOT, ROOT, ROOT)
out = {}
for tv in ("9", "10.5"):
    f = f"/tmp/theta_{tv}.npz"
    r = subprocess.run([sys.executable, "-c", child, f], env=dict(os.environ, DTO_THETA_V=tv), capture_output=True, text=True)
    print(tv, r.stdout.strip()[-80:], r.stderr.strip()[-200:])
    out[tv] = np.load(f)
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
for k in "cjh":
    print(k, "one round vs two rounds:", rel(out["10.5"][k], out["9"][k]))
sys.path.insert(0, ROOT + "/oracle")
import dto_oracle as O
p = O.make_scaled_problem(3, 1024, 4, seed=42)
ev_o = O.OracleEvaluator(p)
co = ev_o.eval_constraint(p.Z0); jo = ev_o.eval_constraint_jacobian(p.Z0)
for tv in out:
    print("theta", tv, "vs oracle: cons", rel(out[tv]["c"], co), "jac", rel(out[tv]["j"], jo))
