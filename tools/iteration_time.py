"""One interior-point iteration's worth of callbacks (g, J, H at the same point, device-resident) at 256x2000, with and
without the reuse_forward_sweep option.  The point changes every iteration, as in a solve."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, dto_amd
n, m, N = 256, 4, 2000
prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m, seed=42)
dev = torch.device("cuda", 0)
Z0 = torch.from_numpy(prob.trajectory.vec()).to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
for reuse in (0, 1, 0, 1):
    ev = dto_amd.Evaluator(prob, eval_hessian=True)
    ev.set_option("reuse_forward_sweep", reuse)
    g = torch.empty(ev.shard.cons_len, dtype=torch.float64, device=dev)
    J = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
    H = torch.empty(ev.shard.hess_len, dtype=torch.float64, device=dev)
    mu = torch.ones(ev.n_constraints, dtype=torch.float64, device=dev)
    Zs = [Z0 + 1e-3 * i * torch.ones_like(Z0) for i in range(8)]
    def iteration(Z):
        ev.eval_constraint_dev(Z.data_ptr(), g.data_ptr(), st)
        ev.eval_jacobian_dev(Z.data_ptr(), J.data_ptr(), st)
        ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), H.data_ptr(), st)
    for Z in Zs[:2]:
        iteration(Z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for Z in Zs[2:]:
        iteration(Z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print(f"reuse={reuse}: {dt*1e3:.2f} ms per iteration (g + J + H), finite={bool(torch.isfinite(H).all())}", flush=True)
    ev.close()
    del g, J, H
    torch.cuda.empty_cache()
