#!/usr/bin/env python3
"""Headline benchmark: knot-points/s of eval_constraint_jacobian, 256-state x 2000-knot bilinear
(BASELINE.json `metric`, configs[2] shape; the reference's generator
benchmark/problem_utils.jl:49-77, callback src/solvers/evaluator.jl:368-380).

A "step" is one eval_constraint_jacobian call over the whole (per-GPU) trajectory with Z and the
value vector resident in HBM.  With --gpus N each rank owns a contiguous knot range of an
N*2000-knot problem (weak scaling, no data-path collective: every rank's output is one contiguous
slab of the CSC value vector); `--scaling strong` splits the 2000 knots of the metric
itself over the N ranks instead, and the weak line carries that measurement as its `strong_scaling`
block.  Started WITHOUT torchrun, `--gpus N` launches its N ranks itself (a child
`python -m torch.distributed.run`, before this process touches a GPU) and relays their line.

Prints ONE JSON line on rank 0."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (AMD datasheet). The microarch
# guide lists no f64 MFMA figure; tools/mfma_f64_rate.py measures the issue rate on the box.
HBM_PEAK_GBS = 8000.0
WATCHDOG_S = int(os.environ.get("DTO_BENCH_WATCHDOG_S", "240"))  # limit for the multi-rank report blocks (gather, strong scaling)


def baseline_metric():
    """The headline metric string, verbatim from BASELINE.json."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "knot-points/sec for eval_constraint_jacobian, 256-state\u00d72000-knot bilinear"


def spawn_ranks(n_ranks, argv):
    """`bench.py --gpus N` started by hand or by a driver that does not use torchrun: launch the N ranks as a child
    `python -m torch.distributed.run` and relay their output.  Runs before this process has made any GPU call (counting
    devices does not initialise the runtime on this image); the parent never touches the GPU and exits with the child's code."""
    import socket
    import subprocess
    if "--one-device" not in argv:
        import torch
        have = torch.cuda.device_count()
        if have < n_ranks:
            raise SystemExit(f"bench.py --gpus {n_ranks}: only {have} GPU(s) visible -- refusing to measure fewer devices than asked for "
                             f"(--one-device with --backend gloo rehearses the multi-rank path on one GPU)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # torch.distributed.run's own parser claims --n: hand the state dimension over under its long name
    argv = ["--states" if a == "--n" else a for a in argv]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def _run_workers(cmds, env=None, timeout=600):
    import subprocess
    procs = [subprocess.Popen(c, stdout=subprocess.PIPE, env=env, text=True) for c in cmds]
    knots, dt = 0, 0.0
    for p in procs:
        out, _ = p.communicate(timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError(f"cpu baseline worker exited with {p.returncode}")
        f = out.split()
        knots += int(f[0])
        dt = max(dt, float(f[1]))
    return knots, dt


def build_ref_costmodel():
    """gcc build of the reference-cost-model restatement (oracle/dto_ref_costmodel.c) ON THIS HOST (-march=native must
    match the cores it is timed on, so a binary built elsewhere is not reused); returns the binary's path."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "dto_ref_costmodel.c")
    exe = os.path.join(tempfile.mkdtemp(prefix="dto_ref_"), "dto_ref_costmodel")
    subprocess.run(["gcc", "-O3", "-march=native", "-o", exe, src, "-lm"], check=True)
    return exe


def cpu_baseline(prob, n, m, N, budget_s=20.0):
    """The reference's CPU path timed beside the GPU on the box's host cores, on a bounded sample of the same workload.

    The reference is Julia and cannot run here, so the baseline is the C restatement of its ALGORITHM
    (oracle/dto_ref_costmodel.c: serial walk over the intervals, ForwardDiff-style forward mode over the 2z inputs in
    chunks of 12 through the truncated-Taylor `expv`, bilinear_integrator.jl:81,111-131): W single-threaded processes (W =
    the CPUs this process may use, at most 16 = the box's share for one GPU) each take every W-th interval of the same synthetic
    problem for `budget_s`/2 seconds; the same on ONE core; and, for context, the repository's own oracle (scipy
    expm + expm_frechet per knot, a different and much cheaper algorithm) on W cores.  Knots are independent, so the
    W-process figure is the CPU's parallel rate."""
    import tempfile
    import numpy as np
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    W = max(1, min(16, avail, N - 1))  # 16 = the box's CPU share for one GPU, whatever the affinity mask shows
    exe = build_ref_costmodel()
    G = prob.integrators[0].G
    z = prob.trajectory.dim
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        np.array([n, m, N, z], dtype=np.int64).tofile(f)
        np.ascontiguousarray(np.transpose(G, (0, 2, 1))).tofile(f)  # column-major per generator
        prob.trajectory.vec()[:N * z].tofile(f)
        path = f.name
    try:
        k_all, t_all = _run_workers([[exe, "bench", path, str(w), str(W), str(budget_s / 2)] for w in range(W)])
        k_one, t_one = _run_workers([[exe, "bench", path, "0", "1", str(budget_s / 4)]])
    finally:
        os.unlink(path)
    out = {"value": k_all / t_all, "unit": "knot-points/s", "cores": W, "kind": "port",
           "sample": f"reference-algorithm restatement in C (forward-mode duals in chunks of 12 through truncated-Taylor expv, "
                     f"{-(-2 * z // 12)} chunks per knot): {W} single-threaded processes over {k_all} of the {N - 1} intervals of "
                     f"the same problem in {t_all:.1f} s",
           "one_core": {"value": k_one / t_one, "knots": k_one, "seconds": t_one}}
    try:
        worker = os.path.join(ROOT, "oracle", "cpu_baseline_worker.py")
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        Wo = min(W, 16)
        k_o, t_o = _run_workers([[sys.executable, worker, ROOT, str(n), str(m), str(N), str(w), str(Wo), str(budget_s / 4)]
                                 for w in range(Wo)], env=env)
        out["scipy_oracle"] = {"value": k_o / t_o, "cores": Wo, "knots": k_o, "seconds": t_o,
                               "note": "oracle/dto_oracle.py (scipy expm + expm_frechet): not the reference's algorithm"}
    except Exception as e:
        out["scipy_oracle"] = {"value": None, "note": f"failed: {e!r}"}
    return out


def other_callbacks(dto_amd, torch, prob, ev_jac, dev, Z, stream, N):
    out = {}
    ev = dto_amd.Evaluator(prob, eval_hessian=True, device=dev.index)
    try:
        mu = torch.ones(ev.n_constraints, dtype=torch.float64, device=dev)
        bufs = {"eval_hessian_lagrangian": torch.empty(ev.shard.hess_len, dtype=torch.float64, device=dev),
                "eval_constraint": torch.empty(ev.shard.cons_len, dtype=torch.float64, device=dev),
                "eval_objective_gradient": torch.empty(ev.shard.grad_len, dtype=torch.float64, device=dev)}
        calls = {"eval_hessian_lagrangian": lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), bufs["eval_hessian_lagrangian"].data_ptr(), stream),
                 "eval_constraint": lambda: ev.eval_constraint_dev(Z.data_ptr(), bufs["eval_constraint"].data_ptr(), stream),
                 "eval_objective_gradient": lambda: ev.eval_gradient_dev(Z.data_ptr(), bufs["eval_objective_gradient"].data_ptr(), stream)}
        for name, fn in calls.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / 5
            out[name] = {"ms_per_call": dt * 1e3, "knot_points_per_s": N / dt,
                         "finite": bool(torch.isfinite(bufs[name]).all().item())}
        # the Hessian's own roofline (configs[2] names eval_hessian_lagrangian): its dominant kernel is the adjoint generator
        # sweep, timed with HIP events in a serial pass (overlap_sweep = 0); the rest of the callback is HBM-bound assembly
        ev.set_option("overlap_sweep", 0)
        ev.profile_enable(True)
        calls["eval_hessian_lagrangian"]()
        torch.cuda.synchronize(dev)
        ev.profile_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            calls["eval_hessian_lagrangian"]()
        torch.cuda.synchronize(dev)
        dt_serial = (time.perf_counter() - t0) / 5
        ev.profile_enable(False)
        ms_a, n_a, fl_a = ev.profile_get("expmv_adjoint")
        nbytes = 8.0 * (ev.shard.hess_len + Z.numel() + ev.n_constraints + sum(it.G.size for it in prob.integrators if hasattr(it, "G")))
        h = out["eval_hessian_lagrangian"]
        if n_a and ms_a > 0:
            ach = fl_a / (ms_a * 1e-3) / 1e12
            h["roofline"] = {"bound": "mfma", "kernel": "k_sweep_fused (adjoint generator sweep: exp(A')mu and its u-tangents)",
                             "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                             "launches": n_a, "avg_launch_ms": ms_a / n_a, "flops_per_launch": fl_a / n_a,
                             "share_of_call": ms_a / n_a / (dt_serial * 1e3), "ms_per_call_serial_pass": dt_serial * 1e3,
                             "measured_in": "serial pass (overlap_sweep = 0), HIP events on the launch stream"}
        h["callback_hbm"] = {"algorithmic_bytes": nbytes, "achieved": nbytes / (h["ms_per_call"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": nbytes / (h["ms_per_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # What Ipopt itself sees: the host-pointer entry points (dto_eval_jacobian / dto_eval_hessian: Z and mu from host memory, the
        # value vector into the solver's pageable host vector -- PCIe in both directions inside the call).  Never part of `value`.
        try:
            import numpy as np
            Zn = prob.trajectory.vec()
            mun = np.ones(ev.n_constraints)
            hp = {}
            jac_h = np.empty(ev.n_jacobian_entries)
            hes_h = np.empty(ev.n_hessian_entries)
            for name, fn, nbytes in (("eval_constraint_jacobian", lambda: ev.eval_constraint_jacobian(jac_h, Zn), jac_h.nbytes),
                                     ("eval_hessian_lagrangian", lambda: ev.eval_hessian_lagrangian(hes_h, Zn, 1.0, mun), hes_h.nbytes)):
                fn()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                hp[name] = {"ms_per_call_median": ts[1] * 1e3, "knot_points_per_s": N / ts[1], "value_vector_bytes": nbytes,
                            "calls_timed": 3}
            hp["note"] = ("host pointers in and out, PCIe included: only the entries that can change cross PCIe (csrc/dto_hostxfer.*), the -E_k "
                          "blocks while the GPU still computes; the rest of the caller's vector is filled by host threads")
            out["host_pointer"] = hp
            del jac_h, hes_h
        except Exception as e:  # a report, never fatal
            out["host_pointer"] = {"error": repr(e)}
        # One interior-point iteration as Ipopt / MadNLP drive it: constraint, objective gradient, Jacobian and Hessian at a NEW
        # point each time (nothing is carried from one point to the next), without and with option reuse_forward_sweep, under
        # which the callbacks of one point share the forward sweep.  A report, never part of `value`.
        ev.set_option("overlap_sweep", 1)
        jac = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
        points = [Z + 1e-7 * (k + 1) for k in range(7)]
        it = {}
        for reuse in (0, 1):
            ev.set_option("reuse_forward_sweep", reuse)

            def iteration(Zk):
                ev.eval_constraint_dev(Zk.data_ptr(), bufs["eval_constraint"].data_ptr(), stream)
                ev.eval_gradient_dev(Zk.data_ptr(), bufs["eval_objective_gradient"].data_ptr(), stream)
                ev.eval_jacobian_dev(Zk.data_ptr(), jac.data_ptr(), stream)
                ev.eval_hessian_dev(Zk.data_ptr(), 1.0, mu.data_ptr(), bufs["eval_hessian_lagrangian"].data_ptr(), stream)

            for Zk in points[:2]:
                iteration(Zk)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for Zk in points[2:]:
                iteration(Zk)
            torch.cuda.synchronize(dev)
            it[reuse] = (time.perf_counter() - t0) / len(points[2:]) * 1e3
        ev.set_option("reuse_forward_sweep", 0)
        out["solver_iteration"] = {"callbacks": "eval_constraint, eval_objective_gradient, eval_constraint_jacobian, eval_hessian_lagrangian "
                                                "at a new point per iteration, device-resident vectors",
                                   "ms_per_iteration": it[0], "ms_per_iteration_reuse_forward_sweep": it[1],
                                   "finite": bool(torch.isfinite(jac).all().item() and torch.isfinite(bufs["eval_hessian_lagrangian"]).all().item())}
    finally:
        ev.close()
    return out


def exchange_comm_id(dto_amd, dist, rank):
    """The 128 bytes of ncclGetUniqueId from rank 0 to every rank (torch.distributed is plumbing here; any transport does)."""
    box = [dto_amd.Evaluator.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def measure_gather(args, dto_amd, torch, dist, ev, prob, dev, Z, stream, fence, N_total, mu, rank, world):
    """configs[3]'s data path through the ENGINE's collectives (C ABI, include/dto_engine.h "Multi-GPU"; RCCL over xGMI):
    every rank allocates the whole value vector with the padding dto_get_gather_layout prescribes, its engine writes the rank's
    slab straight into its slice, dto_gather_*_dev fills in the others with ONE in-place ncclAllGather.  Times K steps of
    callback + gather and K gathers alone; then the overlapped form (the rank's knots over two handles with a communicator
    each: the first half's slabs travel on a second stream while the second half computes)."""
    capi = dto_amd.capi
    vec = capi.VECTOR_JACOBIAN if args.callback == "jacobian" else capi.VECTOR_HESSIAN
    ev.comm_create(exchange_comm_id(dto_amd, dist, rank), rank, world)
    L = ev.gather_layout(vec)
    gbuf = torch.empty(L.padded_len, dtype=torch.float64, device=dev)
    gbuf[:L.front_pad].zero_()
    gbuf[L.front_pad + L.total:].zero_()
    full = gbuf[L.front_pad:L.front_pad + L.total]
    mine = full[L.own_lo:L.own_lo + L.own_len]
    total, ln = L.total, L.own_len
    if args.callback == "jacobian":
        gstep = lambda: ev.eval_jacobian_dev(Z.data_ptr(), mine.data_ptr(), stream)
    else:
        gstep = lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), mine.data_ptr(), stream)
    gather = lambda: ev.gather_dev(vec, gbuf.data_ptr(), stream)
    for _ in range(max(1, args.warmup)):
        gstep()
        gather()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gstep()
        gather()
    fence()
    t_both = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gather()
    fence()
    t_gather = time.perf_counter() - t0
    t_over, over_diff, over_err = float("nan"), None, None
    try:
        D = dto_amd.distributed
        subs = []
        for a, b in D.split_range(ev.shard.k_lo, ev.shard.k_hi, 2):
            e = dto_amd.Evaluator(prob, eval_hessian=(args.callback == "hessian"), device=dev.index, k_lo=a, k_hi=b)
            e.comm_create(exchange_comm_id(dto_amd, dist, rank), rank, world)
            subs.append((e, e.gather_layout(vec)))
        stride = max(1, total // 65536)
        want = full[::stride].clone()
        side = torch.cuda.Stream(device=dev)
        main_stream = torch.cuda.current_stream(dev)

        def ostep():
            evs = []
            for e, Ls in subs:
                dst = full.data_ptr() + 8 * Ls.own_lo
                if args.callback == "jacobian":
                    e.eval_jacobian_dev(Z.data_ptr(), dst, stream)
                else:
                    e.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), dst, stream)
                done = torch.cuda.Event()
                done.record(main_stream)
                side.wait_event(done)               # the gather of this handle's slabs follows ITS callback only
                e.gather_dev(vec, full.data_ptr(), side.cuda_stream)
                g = torch.cuda.Event()
                g.record(side)
                evs.append(g)
            for g in evs:
                main_stream.wait_event(g)
        full.zero_()
        for _ in range(max(1, args.warmup)):
            ostep()
        fence()
        got = full[::stride]
        over_diff = float(((got - want).abs() / want.abs().clamp_min(1.0)).max().item())
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ostep()
        fence()
        t_over = time.perf_counter() - t0
        for e, _ in subs:
            e.comm_destroy()
            e.close()
    except Exception as e:  # reported, never fatal
        over_err = repr(e)
    tt = torch.tensor([t_both, t_gather, t_over], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_both, t_gather, t_over = (float(x) for x in tt.tolist())
    chk = bool(torch.isfinite(full[::max(1, total // 4096)]).all().item())
    res = {"n_ranks": L.world, "transport": "RCCL through the engine's C ABI (dto_comm_create / dto_gather_%s_dev)" % args.callback,
           "bytes_per_rank_vector": 8.0 * total,
           "collective": ("one in-place ncclAllGather on the padded vector" if L.in_place_all_gather else "one in-place ncclBroadcast per rank (grouped)"),
           "ms_per_step_compute_and_gather": t_both / args.steps * 1e3, "gather_ms": t_gather / args.steps * 1e3,
           "gather_gbs_per_rank_received": 8.0 * (total - ln) / (t_gather / args.steps) / 1e9,
           "knot_points_per_s_with_gather": N_total * args.steps / t_both, "sampled_finite": chk,
           "overlap": "ms_per_step_compute_and_gather: none, the gather follows the callback; ms_per_step_overlapped: each rank's "
                      "knots over two handles, the first half's slabs travel while the second half computes (DESIGN.md section 6)"}
    if args.one_device:
        res["wire"] = "rehearsal on one device: the ranks pose as separate hosts (NCCL_HOSTID) and RCCL uses its socket transport"
    if t_over == t_over:
        res.update({"ms_per_step_overlapped": t_over / args.steps * 1e3, "knot_points_per_s_overlapped": N_total * args.steps / t_over,
                    "overlapped_vs_sequential_max_rel_diff_sampled": over_diff})
    if over_err:
        res["overlapped_error"] = over_err
    ev.comm_destroy()
    return res, mine


def measure_strong(args, dto_amd, torch, dist, dev, rank, world, fence, n, m):
    """The literal BASELINE metric at N GPUs: the `--knots`-knot problem itself split over the ranks (strong scaling)."""
    Nk = args.knots
    prob = dto_amd.host.synthetic.make_scaled_problem(Nk, n, m, seed=42)
    lo, hi = dto_amd.distributed.shard_ranges(Nk, world)[rank]
    ev = dto_amd.Evaluator(prob, eval_hessian=False, device=dev.index, k_lo=lo, k_hi=hi)
    try:
        Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
        out = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        step = lambda: ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), stream)
        for _ in range(max(1, args.warmup)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        return {"scaling": "strong", "knots_total": Nk, "knots_per_gpu": hi - lo + 1, "ms_per_step": el / args.steps * 1e3,
                "value": Nk * args.steps / el, "unit": "knot-points/s",
                "outputs_finite": bool(torch.isfinite(out[::max(1, out.numel() // 4096)]).all().item())}
    finally:
        ev.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--states", dest="n", type=int, default=256,
                    help="state dimension (--states under torch.distributed.run, whose parser takes --n for its own)")
    ap.add_argument("--m", type=int, default=4, help="number of drives")
    ap.add_argument("--knots", type=int, default=2000, help="knots per GPU")
    ap.add_argument("--callback", default="jacobian", choices=["jacobian", "hessian", "constraint"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-callbacks", action="store_true")
    ap.add_argument("--no-bound-output", action="store_true", help="skip the bound-output report block (profiling passes: every traced call is then a default call)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="leave the engine's HIP-event kernel timing off (measures its cost; the roofline block is then empty)")
    ap.add_argument("--serial-kernels", action="store_true",
                    help="option overlap_sweep = 0 for the whole run: the generator sweep no longer shares the chip with the chain's "
                         "products, per-kernel durations are clean (the command to put under rocprofv3)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo lets several ranks share one GPU in rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="configs[3] data path: after the compute-only timing, time K more steps in which every rank's slab is "
                         "written into the full value vector and all-gathered in place (RCCL over xGMI); reported next to `value`. "
                         "On by default when there is more than one rank")
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --knots per GPU (configs[3] at 8 GPUs: 16000 knots); strong: --knots in total, split over the GPUs "
                         "(the metric's own 2000-knot problem at 1, 2, 4, 8 GPUs).  The weak line also carries the strong "
                         "measurement as its `strong_scaling` block")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torchrun: start the ranks ourselves, BEFORE this process makes any GPU call, and relay their line
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.one_device and world > 1:
        # one device, several RCCL ranks (rehearsal): every rank poses as a host of its own, RCCL then takes its socket
        # transport over loopback instead of refusing the duplicate device.  Must be in place before RCCL is loaded.
        os.environ["NCCL_HOSTID"] = f"dto-bench-rank-{rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")

    import numpy as np
    import torch
    import dto_amd

    if not args.one_device and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    dist = None
    if args.one_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n, m, Nk = args.n, args.m, args.knots
    strong = args.scaling == "strong"
    N_total = Nk if strong else Nk * world
    prob = dto_amd.host.synthetic.make_scaled_problem(N_total, n, m, seed=42)
    k_lo, k_hi = dto_amd.distributed.shard_ranges(N_total, world)[rank]
    if strong:
        Nk = k_hi - k_lo + 1   # this rank's share (the per-kernel accounting below is per rank)
    ev = dto_amd.Evaluator(prob, eval_hessian=(args.callback == "hessian"), device=local_rank, k_lo=k_lo, k_hi=k_hi)
    Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    sh = ev.shard
    if args.callback == "jacobian":
        out = torch.empty(sh.jac_len, dtype=torch.float64, device=dev)
        step = lambda: ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), stream)
    elif args.callback == "hessian":
        out = torch.empty(sh.hess_len, dtype=torch.float64, device=dev)
        mu = torch.ones(ev.n_constraints, dtype=torch.float64, device=dev)
        step = lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), out.data_ptr(), stream)
    else:
        out = torch.empty(sh.cons_len, dtype=torch.float64, device=dev)
        step = lambda: ev.eval_constraint_dev(Z.data_ptr(), out.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.serial_kernels:
        ev.set_option("overlap_sweep", 0)
    ev.profile_enable(not args.no_kernel_timing)  # warm-up runs with the timing events on too; they are recycled by profile_reset
    for _ in range(args.warmup):
        step()
    fence()
    ev.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ev.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the dominant kernel of the callback: the propagator chain's batched GEMM (Jacobian), the adjoint generator sweep (Hessian:
    # one persistent k_sweep_fused launch, 3.7 of 5.7 ms at 256 x 2000)
    dominant = {"hessian": "expmv_adjoint", "constraint": "expmv"}.get(args.callback, "bgemm")

    n_int = max(0, min(k_hi, N_total - 1) - k_lo + 1)   # intervals this rank owns

    def collect():
        """HIP-event figures of the engine's kernels since the last profile_reset."""
        ms_g, n_g, fl_g = ev.profile_get(dominant)
        # launches of one kind per call = chain chunks (one unless the workspace budget splits the intervals)
        chunks = max(1, round(ev.profile_get("build_A")[1] / max(1, args.steps)))
        ms_s, n_s, fl_s = ev.profile_get("expmv")
        var = {}
        for key, nm in (("horner", "bgemm_horner"), ("square", "bgemm_square"), ("plain", "bgemm_plain"), ("chain64", "chain64"), ("basis", "basis")):
            ms_v, n_v, fl_v = ev.profile_get(nm)
            if n_v:
                var[key] = {"launches": n_v, "avg_launch_ms": ms_v / n_v, "tflops": fl_v / (ms_v * 1e-3) / 1e12}
                # the polynomial products ("horner") also sit on the HBM roof: per launch two operands, three or four power
                # matrices in the epilogue and one or two outputs -- seven npad x npad matrices per interval in each of the
                # three launches of the order-26 form; a squaring moves two
                streams = {"horner": 7, "square": 2}.get(key)
                if streams:
                    npad = -(-n // 64) * 64
                    per_launch = streams * 8.0 * npad * npad * max(n_int, 1) / max(1, chunks)
                    gbs = per_launch / (ms_v / n_v * 1e-3) / 1e9
                    var[key].update({"algorithmic_hbm_bytes": per_launch, "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
        # the bandwidth-bound assembly kernels: achieved HBM GB/s from HIP events and their algorithmic bytes (north_star:
        # "achieved HBM GB/s for the bandwidth-bound assembly")
        asm = {}
        for key, nm, what in (("zero_fill", "zero_fill", "fill!(., 0) of the value slab (k_jac_zero skips the -E_k blocks the chain overwrites; the Hessian's is a memset)"),
                              ("build_A", "build_A", "A_k = dt (G_0 + sum_j u_j G_j): one matrix written per interval (k_build_A)"),
                              ("basis_multi", "basis_multi", "A^2, A^3, A^4 from the generator subspace in one launch: three matrices written per interval (k_basis_gemm_multi)"),
                              ("tangent_columns", "assembly", "tangent columns and identity blocks of the bilinear Jacobian (k_jac_bilinear)")):
            ms_v, n_v, by_v = ev.profile_get(nm)
            if not n_v or ms_v <= 0:
                continue
            if key == "basis_multi":  # the engine prices this launch in flops; its bytes: three output matrices per interval
                npad = -(-n // 64) * 64
                by_v = 3 * 8.0 * npad * npad * n_int * args.steps
            gbs = by_v / (ms_v * 1e-3) / 1e9
            asm[key] = {"kernel": what, "launches": n_v, "avg_launch_ms": ms_v / n_v, "algorithmic_bytes_per_launch": by_v / n_v,
                        "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        return ms_g, n_g, fl_g, ms_s, n_s, fl_s, var, asm

    ms_gemm, n_gemm, fl_gemm, ms_sweep, n_sweep, fl_sweep, variants, assembly = collect()
    # In the timed region the Jacobian's generator sweep shares the chip with the chain's products (option overlap_sweep, on
    # by default): a kernel's HIP-event duration there includes what it waited for the other stream.  The roofline of the
    # dominant kernel is therefore taken from a SERIAL pass -- the same K steps again with overlap_sweep = 0, one kernel at
    # a time, which is also what `rocprofv3 -- python3 bench.py --serial-kernels` shows (profiles/) -- and the timed
    # region's own figures are reported next to it.
    overlapped = None
    if args.callback in ("jacobian", "hessian") and not args.serial_kernels and not args.no_kernel_timing:
        overlapped = {"avg_launch_ms": ms_gemm / max(n_gemm, 1), "launches": n_gemm,
                      "achieved": fl_gemm / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0,
                      "sweep_ms_per_step": ms_sweep / args.steps,
                      "note": ("the chain's launches of the timed region: the sweep runs next to them on a second stream" if args.callback == "jacobian"
                               else "the adjoint sweep of the timed region: the forward sweep of the p column runs next to it on a second stream")}
        ev.set_option("overlap_sweep", 0)
        ev.profile_enable(True)
        step()
        fence()
        ev.profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        serial_elapsed = time.perf_counter() - t0
        ev.profile_enable(False)
        ev.set_option("overlap_sweep", 1)
        ms_gemm, n_gemm, fl_gemm, ms_sweep, n_sweep, fl_sweep, variants, assembly = collect()
        overlapped["ms_per_step_serial_pass"] = serial_elapsed / args.steps * 1e3
    smax, terms = ev.last_stats()
    finite = bool(torch.isfinite(out).all().item())
    if args.callback == "constraint" and n_gemm:
        # the engine prices a one-launch sweep by its step BUDGET; the roofline takes the products of the Taylor terms actually run:
        # 2 npad^2 (m + 1) flops per interval and term (terms - 1 products: term 0 is the state itself)
        npad = -(-n // 64) * 64
        fl_gemm = 2.0 * npad * npad * (m + 1) * n_int * max(terms - 1, 1) * n_gemm

    # The blocks below are reports next to `value` (gather, strong scaling).  With several ranks they run collectives that
    # have never executed on an 8-GPU node: should one of them hang, every rank leaves after WATCHDOG_S seconds and rank 0
    # still prints the headline measured above (marked as such) instead of losing the line.
    watchdog = None
    if dist is not None and world > 1:
        import threading

        def bail():
            if rank == 0:
                print(json.dumps({
                    "metric": baseline_metric() if (args.callback, n, args.knots) == ("jacobian", 256, 2000)
                    else f"knot-points/sec for eval_{args.callback}, {n}-state x {args.knots}-knot bilinear",
                    "value": N_total * args.steps / elapsed, "unit": "knot-points/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                    "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                    "config": {"workload": f"configs[2]: {n}-state bilinear, {m} drives, N={N_total} knots over {world} GPU(s), "
                                           f"callback={args.callback}", "knots_per_gpu": Nk, "knots_total": N_total,
                               "parallelism": f"knot-range shards x{world}"},
                    "status": "collective_timeout",
                    "watchdog": f"the gather / strong-scaling reports did not finish within {WATCHDOG_S} s; `value` is the "
                                "compute-only headline measured before them"}), flush=True)
            # the line above keeps the headline, the exit code says that a collective hung: never report success for it
            os._exit(3)

        watchdog = threading.Timer(WATCHDOG_S, bail)
        watchdog.daemon = True
        watchdog.start()

    gather = None
    want_gather = args.gather if args.gather is not None else world > 1
    if want_gather and dist is not None and args.callback in ("jacobian", "hessian"):
        n_out = out.numel()
        del out  # its place is taken by the rank's slice of the full vector
        try:
            gather, out = measure_gather(args, dto_amd, torch, dist, ev, prob, dev, Z, stream, fence, N_total,
                                         mu if args.callback == "hessian" else None, rank, world)
        except Exception as e:  # the gather is a report next to `value`, never a reason to lose the line
            gather, out = {"error": repr(e)}, torch.empty(n_out, dtype=torch.float64, device=dev)

    # the metric's own problem split over the ranks, next to the weak-scaling `value` (all ranks take part)
    strong_block = None
    if dist is not None and not strong and args.callback == "jacobian":
        try:
            strong_block = measure_strong(args, dto_amd, torch, dist, dev, rank, world, fence, n, m)
        except Exception as e:
            strong_block = {"error": repr(e)}

    if watchdog is not None:
        watchdog.cancel()

    # per-step times, each step fenced on its own: the median next to the mean that `value` is (SURVEY.md §8d quotes a median)
    step_ms = []
    if args.callback == "jacobian" and world == 1:
        ostep = (lambda: ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), stream))
        for _ in range(args.steps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ostep()
            torch.cuda.synchronize(dev)
            step_ms.append((time.perf_counter() - t0) * 1e3)

    # ... and the same with the host-to-device copy of Z inside each timed step (SURVEY.md section 8d's metric definition)
    h2d_ms = []
    if args.callback == "jacobian" and world == 1:
        Zh = torch.from_numpy(prob.trajectory.vec()).pin_memory()
        for _ in range(args.steps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            Z.copy_(Zh, non_blocking=True)
            ostep()
            torch.cuda.synchronize(dev)
            h2d_ms.append((time.perf_counter() - t0) * 1e3)

    # The same K steps into a BOUND output vector (dto_bind_output_dev: what a solver in GPU mode, which hands the engine the same
    # device vector every iteration, can declare): the call-invariant half of the slab is then written once, not per call.
    # Reported next to `value`, never as `value` -- the headline keeps the reference's semantics (every entry written per call).
    bound_block = None
    if args.callback in ("jacobian", "hessian") and world == 1 and not args.no_bound_output:
        vec = dto_amd.capi.VECTOR_JACOBIAN if args.callback == "jacobian" else dto_amd.capi.VECTOR_HESSIAN
        ev.bind_output_dev(vec, out.data_ptr())
        for _ in range(max(2, args.warmup)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        tb = time.perf_counter() - t0
        ev.bind_output_dev(vec, 0)
        bound_block = {"ms_per_step": tb / args.steps * 1e3, "knot_points_per_s": N_total * args.steps / tb,
                       "note": "dto_bind_output_dev: constants (structural zeros, identity blocks) written once; same buffer every call"}

    traffic, traffic_src, traffic_call = None, None, None
    try:  # HBM bytes per launch of the dominant kernel come from committed PMC passes (bench.py cannot run rocprofv3 on itself)
        import glob
        # the latest round's passes for THIS shape and callback (profiles/rNN*_traffic_<callback>_<n>x<knots>.json), else none
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]*_traffic_{args.callback}_{n}x{Nk}.json")))
        if cands:
            tj = json.load(open(cands[-1]))
            traffic, traffic_src = tj["avg_per_launch_bytes"], tj["source"]
            traffic_call = tj.get("per_call_total_bytes")
            pl = tj["per_launch_bytes"]
            prods = [v["read"] + v["write"] for k, v in pl.items() if k.startswith("product")]
            if "horner" in variants and prods:
                variants["horner"]["traffic"] = sum(prods) / len(prods)
            if "square" in variants and "square" in pl:
                variants["square"]["traffic"] = pl["square"]["read"] + pl["square"]["write"]
    except Exception:
        pass
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        achieved = fl_gemm / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0
        line = {
            "metric": baseline_metric() if (args.callback, n, args.knots) == ("jacobian", 256, 2000)
            else f"knot-points/sec for eval_{args.callback}, {n}-state x {args.knots}-knot bilinear",
            "value": N_total * args.steps / elapsed,
            "unit": "knot-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"configs[2]: {n}-state bilinear, {m} drives, N={N_total} knots split over {world} GPU(s) " if strong else
                                    f"configs[2]: {n}-state bilinear, {m} drives, N={Nk} knots per GPU ") +
                                   f"(make_scaled_problem shape, Philox seed 42), callback={args.callback}",
                       "state_dim": n, "drives": m, "knots_per_gpu": Nk, "knots_total": N_total,
                       "parallelism": f"knot-range shards x{world}", "outputs_finite": finite,
                       "max_squarings": smax, "sweep_terms": terms,
                       "inputs": "Z and the value slab resident in HBM; the 4.2 MB host-to-device copy of Z that SURVEY.md §8d's "
                                 "metric lists is NOT in `value` (host-pointer figures: DESIGN.md §5)"},
            "roofline": {
                "bound": "mfma", "kernel": {"hessian": ("k_sweep_s64" if n <= 64 else "k_sweep_fused / k_sweep_gs") +
                                                       " (adjoint generator sweep of the Hessian: exp(A')mu and its u-tangents, one persistent launch)",
                                            "constraint": ("k_sweep_s64" if 32 < n <= 64 else "k_sweep_gs / k_sweep") + " (generator sweep of the p column: exp(A)x)"}.get(
                                                args.callback, "k_chain64 (the propagator chain of a 33..64-state integrator in ONE launch, a workgroup per interval: "
                                                               "powers, polynomial products and squarings on FP64 MFMA out of LDS; priced at six 64^3 products per interval)"
                                                if "chain64" in variants else "k_bgemm (batched FP64 MFMA GEMM of the propagator chain)"),
                "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                # HBM-side bytes of ALL kernels of one call from the same PMC passes, against SURVEY.md §8d's algorithmic bytes
                "traffic_per_call": traffic_call,
                "launches": n_gemm, "avg_launch_ms": ms_gemm / max(n_gemm, 1),
                "flops_per_launch": fl_gemm / max(n_gemm, 1),
                "share_of_step": ms_gemm / ((overlapped["ms_per_step_serial_pass"] if overlapped else ms_per_step) * args.steps)
                if ms_per_step > 0 else None,
                "template_instances": variants,
                "measured_in": ("serial pass: the K steps again with option overlap_sweep = 0 (one kernel at a time), HIP events "
                                "on the launch stream; same as rocprofv3 of `bench.py --serial-kernels`" if overlapped else
                                "the timed region"),
                "timed_region": overlapped,
            },
            # the whole callback against the HBM roofline (BASELINE north_star asks for this fraction too): algorithmic
            # bytes = mandatory output write + read of Z + one read of the generators
            "callback_hbm": (lambda nbytes: {"algorithmic_bytes": nbytes, "achieved": nbytes / (ms_per_step * 1e-3) / 1e9,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": nbytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS})(
                8.0 * (out.numel() + Z.numel() + (m + 1) * n * n + (ev.n_constraints if args.callback == "hessian" else 0))),
            # one timed region per sweep (its step kernels, termination tests and the gaps between them); flops = the
            # generator products of the Taylor terms actually used: 2 npad^2 (m+1) products per column, (1+m) column types
            # (Jacobian; the Hessian's two sweeps and the constraint's single column are priced by the engine's own count)
            "secondary_kernel": (lambda fl: {"kernel": "k_sweep (generator sweep: exp(A)x and its u-tangents)",
                                             "ms_per_step": ms_sweep / args.steps, "timed_regions": n_sweep,
                                             "achieved_tflops": fl / (ms_sweep * 1e-3) / 1e12 if ms_sweep > 0 else 0.0})(
                (2.0 * (-(-n // 64) * 64) ** 2 * (-(-(Nk - 1) // 128) * 128) * (m + 1) * (m + 1) * terms * args.steps)
                if args.callback == "jacobian" else fl_sweep),
        }
        if assembly:
            line["assembly_hbm"] = assembly
        if h2d_ms:
            hs = sorted(h2d_ms)
            med = hs[len(hs) // 2] if len(hs) % 2 else 0.5 * (hs[len(hs) // 2 - 1] + hs[len(hs) // 2])
            # SURVEY.md section 8d's literal definition: median wall time of one full callback INCLUDING the host-to-device copy of Z
            line["value_incl_h2d_median"] = {"value": N_total / (med * 1e-3), "unit": "knot-points/s", "ms_per_step_median": med,
                                             "includes": "H2D copy of Z from pinned host memory on the call's stream + the callback; "
                                                         "values stay resident in HBM (D2H reported under host_pointer)"}
        if step_ms:
            ss = sorted(step_ms)
            line["ms_per_step_median"] = ss[len(ss) // 2] if len(ss) % 2 else 0.5 * (ss[len(ss) // 2 - 1] + ss[len(ss) // 2])
            line["value_from_median"] = N_total / (line["ms_per_step_median"] * 1e-3)
        if bound_block is not None:
            line["bound_output"] = bound_block
        if gather is not None:
            line["gather"] = gather
        if strong_block is not None:
            line["strong_scaling"] = strong_block
        if world == 1 and args.callback == "jacobian" and not args.no_other_callbacks:
            # the other callbacks of the same problem, same protocol (3 untimed + 5 timed calls each): reported for
            # context (SURVEY.md §8d lists them next to the headline), never part of `value`
            line["other_callbacks"] = other_callbacks(dto_amd, torch, prob, ev, dev, Z, stream, N_total)
            if "host_pointer" in line["other_callbacks"]:
                line["host_pointer"] = line["other_callbacks"].pop("host_pointer")
        if world == 1 and not args.no_cpu_baseline and args.callback == "jacobian":
            try:
                line["cpu_baseline"] = cpu_baseline(prob, n, m, Nk, args.cpu_budget)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "knot-points/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    ev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
