"""Jacobian time of the 256-state x 2000-knot problem at smaller time steps (the headline uses dt = 0.1, alpha ~ 4.1):
inside the radius of a polynomial form the propagator chain runs no squaring at all."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, dto_amd
prob = dto_amd.host.synthetic.make_scaled_problem(2000, 256, 4, seed=42)
ev = dto_amd.Evaluator(prob, eval_hessian=False)
dev = torch.device("cuda", 0)
Z0 = prob.trajectory.vec()
st = torch.cuda.current_stream(dev).cuda_stream
out = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
z = 256 + 2 * 4 + 1
for dt in (0.1, 0.06, 0.03, 0.015):
    Zh = Z0.copy(); Zh[z - 1::z][:2000] = dt
    Z = torch.from_numpy(Zh).to(dev)
    f = lambda: ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(f"dt={dt}: {t*1e3:.2f} ms per Jacobian, stats={ev.last_stats()} finite={bool(torch.isfinite(out).all())}", flush=True)
ev.close()
