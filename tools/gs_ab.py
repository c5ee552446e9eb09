#!/usr/bin/env python3
"""A/B of the generator-stationary sweep inside one process (TUNING build: DTO_ENGINE_LIB=libdto_engine_t.so, DTO_SWEEP_GS=0/1 is
read when the first handle plans a sweep, so each arm is its own process): callbacks at the shapes the form serves."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dto_amd

dev = torch.device("cuda", 0)
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(256, 2000), (256, 250), (256, 500)]
for n, N in SHAPES:
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, 4, seed=42)
    Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ev = dto_amd.Evaluator(prob, eval_hessian=True)
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
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        out.append(f"{name} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
    print(f"{n} x {N} DTO_SWEEP_GS={os.environ.get('DTO_SWEEP_GS', 'default')}: " + ", ".join(out), flush=True)
    ev.close()
