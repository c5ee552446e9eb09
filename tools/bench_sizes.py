"""Callback times at other shapes than the headline's, with the fused sweep and the step-per-launch sweep (option sweep_form)
side by side.  usage: python tools/bench_sizes.py [--big] [--tdb-only] [--host-mirror]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dto_amd


def run(n, m, N, cb="jacobian", steps=5, make=None):
    prob = (make or dto_amd.host.synthetic.make_scaled_problem)(N, n, m, seed=42)
    ev = dto_amd.Evaluator(prob, eval_hessian=(cb == "hessian"))
    dev = torch.device("cuda", 0)
    Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if cb == "jacobian":
        out = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev); f = lambda: ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
    elif cb == "hessian":
        out = torch.empty(ev.shard.hess_len, dtype=torch.float64, device=dev); mu = torch.ones(ev.n_constraints, dtype=torch.float64, device=dev)
        f = lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), out.data_ptr(), st)
    else:
        out = torch.empty(ev.shard.cons_len, dtype=torch.float64, device=dev); f = lambda: ev.eval_constraint_dev(Z.data_ptr(), out.data_ptr(), st)
    res = []
    for form in (0, 1):
        ev.set_option("sweep_form", form)
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        res.append(dt * 1e3)
    print(f"n={n} m={m} N={N} {cb}: fused {res[0]:.3f} ms, step-per-launch {res[1]:.3f} ms  ({N / (res[0] * 1e-3):.0f} knot-points/s) "
          f"finite={bool(torch.isfinite(out).all())} stats={ev.last_stats()}", flush=True)
    ev.close()


TDB_ONLY = "--tdb-only" in sys.argv
for cb in (() if TDB_ONLY else ("constraint", "jacobian", "hessian")):
    run(64, 4, 1000, cb)
for n, N in () if TDB_ONLY else ((128, 1000), (256, 200), (256, 2000), (192, 500), (512, 500)):
    for cb in ("jacobian", "hessian"):
        run(n, 4, N, cb, steps=3)
if "--big" in sys.argv:
    for cb in ("constraint", "jacobian", "hessian"):
        run(1024, 4, 500, cb, steps=2, make=dto_amd.host.synthetic.make_l1_slack_problem)  # configs[4] per-GPU share: N=4000 over 8 GPUs
    for cb in ("jacobian", "hessian"):
        run(256, 4, 16000, cb, steps=2)  # configs[3] unsharded
if not TDB_ONLY:
    run(4, 2, 51, "jacobian"); run(4, 2, 51, "hessian")
    for cb in ("constraint", "jacobian", "hessian"):
        run(32, 4, 100, cb)


def run_tdb(n, m, N, order=1, substeps=32, steps=3):
    """TimeDependentBilinearIntegrator on the device (csrc/dto_tdb.hip) against the host-evaluated
    merge path (Python RK4 + complex-step / difference derivatives -- what a closure G(u, t) costs)."""
    rng = np.random.default_rng(5)
    traj = dto_amd.NamedTrajectory({"x": rng.standard_normal((n, N)), "u": 0.4 * rng.standard_normal((m, N)),
                                    "t": np.cumsum(np.full(N, 0.3))[None, :], "dt": np.full((1, N), 0.25)}, timestep="dt")
    fam = dto_amd.ModulatedGenerators(rng.standard_normal((m + 1, n, n)) / np.sqrt(n),
                                      [("cos", 1.7, 0.5 * rng.standard_normal((m + 1, n, n)) / np.sqrt(n))])
    Z = traj.vec()
    for on_device in (True, False):
        if not on_device and (n * N > 400 or "--host-mirror" not in sys.argv):
            continue  # the host mirror needs 93 s for ONE Hessian at n = 4, N = 100: only on request
        tdb = dto_amd.TimeDependentBilinearIntegrator(fam, "x", "u", "t", traj, spline_order=order, substeps=substeps, on_device=on_device)
        ev = dto_amd.Evaluator(dto_amd.DirectTrajOptProblem(traj, dto_amd.QuadraticRegularizer("u", traj, 1.0), [tdb]))
        mu = np.ones(ev.n_constraints)
        out = {}
        for name, fn in (("constraint", lambda: ev.eval_constraint(np.empty(ev.n_constraints), Z)),
                         ("jacobian", lambda: ev.eval_constraint_jacobian(np.empty(ev.n_jacobian_entries), Z)),
                         ("hessian", lambda: ev.eval_hessian_lagrangian(np.empty(ev.n_hessian_entries), Z, 1.0, mu))):
            fn()
            t0 = time.perf_counter()
            for _ in range(steps if on_device else 1):
                fn()
            out[name] = (time.perf_counter() - t0) / (steps if on_device else 1) * 1e3
        print(f"time-dependent bilinear n={n} m={m} N={N} order={order} substeps={substeps} {'device' if on_device else 'host mirror'}: "
              + ", ".join(f"{k} {v:.2f} ms" for k, v in out.items()) + " (host-pointer calls)", flush=True)
        ev.close()


run_tdb(4, 2, 100)
run_tdb(4, 2, 1000)
run_tdb(16, 2, 1000)
run_tdb(32, 4, 500)
