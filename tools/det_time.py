#!/usr/bin/env python3
"""Jacobian at 256 x 2000 with and without option `deterministic` (the sweep beside the chain in the shape it has alone -- K split,
223 workgroups -- instead of the two-column-group shape, 167 workgroups).  usage: python tools/det_time.py [knots=2000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dto_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
prob = dto_amd.host.synthetic.make_scaled_problem(N, 256, 4, seed=42)
ev = dto_amd.Evaluator(prob, eval_hessian=False)
dev = torch.device("cuda", 0)
Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
out = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for rep in range(3):
    for det in (0, 1):
        ev.set_option("deterministic", det)
        for _ in range(3):
            ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
        torch.cuda.synchronize()
        print(f"deterministic={det}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
ev.close()
