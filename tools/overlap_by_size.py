"""Jacobian time with and without option overlap_sweep at shapes whose sweep takes the step-per-launch form or a small fused form
(the headline shape is A/B-ed by tools/ab_env.sh).  usage: python tools/overlap_by_size.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dto_amd
dev = torch.device("cuda", 0)
for n, N in ((512, 500), (1024, 300), (256, 200), (192, 500), (320, 800)):
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, 4, seed=42)
    ev = dto_amd.Evaluator(prob, eval_hessian=False)
    Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    out = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
    res = {}
    for rep in range(2):
        for on in (1, 0):
            ev.set_option("overlap_sweep", on)
            for _ in range(2): ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
            torch.cuda.synchronize(); res.setdefault(on, []).append((time.perf_counter() - t0) / 4 * 1e3)
    print(f"n={n} N={N}: overlap on {min(res[1]):.3f} ms, off {min(res[0]):.3f} ms", flush=True)
    ev.close()
