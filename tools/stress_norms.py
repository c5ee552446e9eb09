import sys, os
ROOT='/root/repo'
for p in (ROOT, ROOT+'/oracle', ROOT+'/tests'): sys.path.insert(0,p)
import numpy as np, dto_amd, dto_oracle as O
from helpers import to_engine, rel_err, run_all
def relmax(a,b):
    a=np.asarray(a); b=np.asarray(b); return float(np.max(np.abs(a-b))/max(1.0,np.max(np.abs(b))))
for (n,m,N,dtv,skew) in [(64,4,4,1.0,False),(32,2,4,3.0,False),(24,3,4,2.0,True),(128,2,3,0.5,False),(20,2,4,1e-8,False)]:
    p = O.make_scaled_problem(N, n, m, seed=n, skew=skew)
    Z = p.Z0.copy(); Z[p.dt_idx::p.z] = dtv
    ev_o = O.OracleEvaluator(p); ev = dto_amd.Evaluator(to_engine(p))
    mu = np.random.default_rng(1).standard_normal(ev_o.n_constraints)
    out = run_all(ev, p, Z, mu, sigma=1.0)
    ref_j = ev_o.eval_constraint_jacobian(Z); ref_h = ev_o.eval_hessian_lagrangian(Z,1.0,mu); ref_c = ev_o.eval_constraint(Z)
    print(f"n={n} dt={dtv} skew={skew}: |J|max={np.abs(ref_j).max():.3e} relmax err cons {relmax(out['cons'],ref_c):.2e} jac {relmax(out['jac'],ref_j):.2e} hess {relmax(out['hess'],ref_h):.2e} stats={ev.last_stats()}", flush=True)
    ev.close()
