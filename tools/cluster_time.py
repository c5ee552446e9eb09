#!/usr/bin/env python3
"""Timing of the callbacks the cluster sweep serves (a tools/ probe): eval_constraint / Jacobian / Hessian at 256 x 2000 and at the
250-knot strong-scaling share, cluster form against step-per-launch (option sweep_form = 1)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dto_amd

dev = torch.device("cuda", 0)
import sys as _s
SHAPES = [tuple(int(v) for v in a.split("x")) for a in _s.argv[1:]] or [(256, 2000), (256, 250), (128, 500), (256, 500)]
for n, N in SHAPES:
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, 4, seed=42)
    Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for form in (0, 1):
        ev = dto_amd.Evaluator(prob, eval_hessian=True)
        ev.set_option("sweep_form", form)
        mu = torch.ones(ev.n_constraints, dtype=torch.float64, device=dev)
        bufs = {"cons": torch.empty(ev.shard.cons_len, dtype=torch.float64, device=dev),
                "jac": torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev),
                "hess": torch.empty(ev.shard.hess_len, dtype=torch.float64, device=dev)}
        calls = {"cons": lambda: ev.eval_constraint_dev(Z.data_ptr(), bufs["cons"].data_ptr(), st),
                 "jac": lambda: ev.eval_jacobian_dev(Z.data_ptr(), bufs["jac"].data_ptr(), st),
                 "hess": lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), bufs["hess"].data_ptr(), st)}
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
        print(f"{n} x {N} sweep_form={form}: " + ", ".join(out), "finite", bool(torch.isfinite(bufs['jac']).all()), flush=True)
        ev.close()
