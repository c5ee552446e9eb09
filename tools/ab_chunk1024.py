"""1024 x 500 Jacobian / Hessian / constraint: the chain (and exact-norm passes) in one chunk against two (option chain_chunk), same process.
usage (through gpurun): python tools/ab_chunk1024.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dto_amd

dev = torch.device("cuda", 0)
p = dto_amd.host.synthetic.make_scaled_problem(500, 1024, 4, seed=42)
ev = dto_amd.Evaluator(p, device=0)
Z = torch.from_numpy(p.trajectory.vec()).to(dev)
mu = torch.ones(ev.n_constraints, dtype=torch.float64, device=dev)
J = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
H = torch.empty(ev.shard.hess_len, dtype=torch.float64, device=dev)
g = torch.empty(ev.n_constraints, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
calls = {"jacobian": lambda: ev.eval_jacobian_dev(Z.data_ptr(), J.data_ptr(), st),
         "hessian": lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), H.data_ptr(), st),
         "constraint": lambda: ev.eval_constraint_dev(Z.data_ptr(), g.data_ptr(), st)}
for rep in range(2):
    for chunk in (0, 256):
        ev.set_option("chain_chunk", chunk)
        for name, fn in calls.items():
            fn(); torch.cuda.synchronize()
            n = 8 if name == "jacobian" else 12
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            print(f"chain_chunk={chunk or 'one chunk'} {name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms", flush=True)
ev.close()
