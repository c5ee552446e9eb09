#!/usr/bin/env python3
"""256 x 2000: eval_constraint_jacobian / eval_hessian_lagrangian into a BOUND device vector against the plain call (a tools/ probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, dto_amd
from dto_amd import capi
dev = torch.device("cuda", 0)
prob = dto_amd.host.synthetic.make_scaled_problem(2000, 256, 4, seed=42)
Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
ev = dto_amd.Evaluator(prob, eval_hessian=True)
mu = torch.ones(ev.n_constraints, dtype=torch.float64, device=dev)
J = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
H = torch.empty(ev.shard.hess_len, dtype=torch.float64, device=dev)
calls = {"jac": lambda: ev.eval_jacobian_dev(Z.data_ptr(), J.data_ptr(), st),
         "hess": lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), H.data_ptr(), st)}
for rep in range(2):
    for bound in (0, 1):
        ev.bind_output_dev(capi.VECTOR_JACOBIAN, J.data_ptr() if bound else 0)
        ev.bind_output_dev(capi.VECTOR_HESSIAN, H.data_ptr() if bound else 0)
        out = []
        for name, fn in calls.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            out.append(f"{name} {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
        print("bound" if bound else "plain", ", ".join(out), flush=True)
