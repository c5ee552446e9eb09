"""PCIe-inclusive time of the host-pointer entry point (Z from host memory, value vector back to host)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, dto_amd
prob = dto_amd.host.synthetic.make_scaled_problem(2000, 256, 4, seed=42)
ev = dto_amd.Evaluator(prob, eval_hessian=True)
Z = prob.trajectory.vec()
vals = np.empty(ev.n_jacobian_entries)
ev.eval_constraint_jacobian(vals, Z)
t0 = time.perf_counter()
for _ in range(3): ev.eval_constraint_jacobian(vals, Z)
dt = (time.perf_counter() - t0) / 3
print(f"host-pointer eval_constraint_jacobian 256x2000: {dt*1e3:.1f} ms ({2000/dt:.0f} knot-points/s), value vector {vals.nbytes/1e9:.2f} GB")
H = np.empty(ev.n_hessian_entries); mu = np.ones(ev.n_constraints)
ev.eval_hessian_lagrangian(H, Z, 1.0, mu)
t0 = time.perf_counter()
for _ in range(3): ev.eval_hessian_lagrangian(H, Z, 1.0, mu)
dt = (time.perf_counter() - t0) / 3
print(f"host-pointer eval_hessian_lagrangian 256x2000: {dt*1e3:.1f} ms ({2000/dt:.0f} knot-points/s), value vector {H.nbytes/1e9:.2f} GB")
w = np.random.default_rng(0).standard_normal(ev.n_variables); y = np.empty(ev.n_constraints)
ev.eval_constraint_jacobian_product(y, Z, w)
t0 = time.perf_counter()
for _ in range(3): ev.eval_constraint_jacobian_product(y, Z, w)
dt = (time.perf_counter() - t0) / 3
print(f"host-pointer J*w (matrix-free) 256x2000: {dt*1e3:.1f} ms")
v = np.random.default_rng(1).standard_normal(ev.n_constraints); yt = np.empty(ev.n_variables)
ev.eval_constraint_jacobian_transpose_product(yt, Z, v)
t0 = time.perf_counter()
for _ in range(3): ev.eval_constraint_jacobian_transpose_product(yt, Z, v)
dt = (time.perf_counter() - t0) / 3
print(f"host-pointer J'*w (matrix-free) 256x2000: {dt*1e3:.1f} ms")
