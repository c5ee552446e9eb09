"""Host-pointer callbacks (what Ipopt calls: host vectors in, host vectors out) at the headline shape: with the variable-run
hand-off (dto_hostxfer.h) and with the plain whole-slab copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, dto_amd
n, m, N = 256, 4, 2000
prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m)
Z = prob.trajectory.vec()
for on in (1, 0):
    ev = dto_amd.Evaluator(prob)
    ev.set_option("host_xfer", on)
    mu = np.ones(ev.n_constraints)
    j = np.zeros(ev.n_jacobian_entries); h = np.zeros(ev.n_hessian_entries)  # touched pages, as a solver's buffers are
    for name, fn in (("eval_constraint_jacobian", lambda: ev.eval_constraint_jacobian(j, Z)),
                     ("eval_hessian_lagrangian", lambda: ev.eval_hessian_lagrangian(h, Z, 1.0, mu))):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        dt = (time.perf_counter() - t0) / 5
        print(f"host_xfer={on} {name}: {dt * 1e3:.1f} ms per call (host pointers, PCIe included)", flush=True)
    ev.close()
