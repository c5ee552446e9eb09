import sys, os
ROOT='/root/repo'
for p in (ROOT, ROOT+'/oracle', ROOT+'/tests'): sys.path.insert(0,p)
import numpy as np, dto_amd, dto_oracle as O
from helpers import to_engine, run_all
def relmax(a,b):
    a=np.asarray(a); b=np.asarray(b); return float(np.max(np.abs(a-b))/max(1.0,np.max(np.abs(b))))
rng=np.random.default_rng(2026)
worst={"cons":0,"jac":0,"hess":0}
for it in range(40):
    n=int(rng.choice([33,40,48,64,70,96,128,130,160,192])); m=int(rng.integers(1,6)); N=int(rng.integers(3,7))
    skew=bool(rng.integers(0,2)); 
    p=O.make_scaled_problem(N,n,m,seed=int(rng.integers(1,10**6)),skew=skew,with_constraint=bool(rng.integers(0,2)))
    Z=p.Z0.copy()
    # time steps spread over two decades so that both polynomial forms, several squaring counts and q > 1 occur
    base=0.1*np.sqrt(256.0/n)*(1.0 if not skew else 4.0)
    Z[p.dt_idx::p.z]=base*10**rng.uniform(-1.0,0.9,p.N)
    ev_o=O.OracleEvaluator(p); ev=dto_amd.Evaluator(to_engine(p))
    mu=rng.standard_normal(ev_o.n_constraints)
    out=run_all(ev,p,Z,mu,sigma=0.8)
    e={"cons":relmax(out["cons"],ev_o.eval_constraint(Z)),"jac":relmax(out["jac"],ev_o.eval_constraint_jacobian(Z)),"hess":relmax(out["hess"],ev_o.eval_hessian_lagrangian(Z,0.8,mu))}
    st=ev.last_stats(); ev.close()
    for k in e: worst[k]=max(worst[k],e[k])
    print(f"{it:2d} n={n} m={m} N={N} skew={skew} stats={st} " + " ".join(f"{k} {v:.1e}" for k,v in e.items()), flush=True)
print("WORST", worst)
