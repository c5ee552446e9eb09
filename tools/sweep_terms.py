"""Distribution of the Taylor terms the fused sweep's workgroups use (diagnostics): runs one Jacobian of the headline
problem and reads the sweep statistics [non-converged, max terms, sum of terms over workgroups]."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dto_amd
n, m, N = (int(a) for a in (sys.argv[1:4] + ["256", "4", "2000"][len(sys.argv) - 1:]))
prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m)
ev = dto_amd.Evaluator(prob, eval_hessian=False)
Z = prob.trajectory.vec()
j = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(j, Z)
print("last_stats (max squarings, max terms):", ev.last_stats())
