#!/usr/bin/env python3
"""Headline benchmark: knot-points/s of eval_constraint_jacobian, 256-state x 2000-knot bilinear
(BASELINE.json `metric`, configs[2] shape; the reference's generator
benchmark/problem_utils.jl:49-77, callback src/solvers/evaluator.jl:368-380).

A "step" is one eval_constraint_jacobian call over the whole (per-GPU) trajectory with Z and the
value vector resident in HBM.  With --gpus N each rank owns a contiguous knot range of an
N*2000-knot problem (weak scaling, no data-path collective: every rank's output is one contiguous
slab of the CSC value vector).

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


def baseline_metric():
    """The headline metric string, verbatim from BASELINE.json."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "knot-points/sec for eval_constraint_jacobian, 256-state\u00d72000-knot bilinear"


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
    finally:
        ev.close()
    return out


def measure_gather(args, dto_amd, torch, dist, ev, prob, dev, Z, stream, fence, N_total, mu):
    """configs[3]'s data path: every rank holds the WHOLE value vector; its engine writes the rank's slab straight into its
    slice, the other slices arrive by the in-place all-gather (dto_amd.distributed.gather_slabs_inplace: RCCL over xGMI
    with backend nccl; no padded copy, no concatenation).  Times K steps of callback + gather and K gathers alone."""
    sh = ev.shard
    lo, ln = (sh.jac_lo, sh.jac_len) if args.callback == "jacobian" else (sh.hess_lo, sh.hess_len)
    total = ev.n_jacobian_entries if args.callback == "jacobian" else ev.n_hessian_entries
    layout = dto_amd.distributed.slab_layout(lo, ln)
    # padded allocation: the first and the last rank's slabs are one boundary half-block shorter than the others; with that
    # much padding in front and behind, ONE equal-size all-gather moves every slab in place (host/distributed.py)
    gbuf, full = dto_amd.distributed.alloc_gather_vector(total, layout, torch.float64, dev)
    mine = full[lo:lo + ln]
    if args.callback == "jacobian":
        gstep = lambda: ev.eval_jacobian_dev(Z.data_ptr(), mine.data_ptr(), stream)
    else:
        gstep = lambda: ev.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), mine.data_ptr(), stream)
    for _ in range(max(1, args.warmup)):
        gstep()
        dto_amd.distributed.gather_slabs_inplace(full, layout, buffer=gbuf)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gstep()
        dto_amd.distributed.gather_slabs_inplace(full, layout, buffer=gbuf)
    fence()
    t_both = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dto_amd.distributed.gather_slabs_inplace(full, layout, buffer=gbuf)
    fence()
    t_gather = time.perf_counter() - t0
    # Overlapped form: the rank's knots over two engine handles; while the second half computes, the first half's slabs are
    # already on the links (asynchronous in-place broadcasts on the backend's stream).  Same collectives in the same order
    # on every rank.
    t_over, over_diff, over_err = float("nan"), None, None
    try:
        D = dto_amd.distributed
        k_lo, k_hi = ev.shard.k_lo, ev.shard.k_hi
        subs = []
        for a, b in D.split_range(k_lo, k_hi, 2):
            e = dto_amd.Evaluator(prob, eval_hessian=(args.callback == "hessian"), device=dev.index, k_lo=a, k_hi=b)
            slo, sln = (e.shard.jac_lo, e.shard.jac_len) if args.callback == "jacobian" else (e.shard.hess_lo, e.shard.hess_len)
            subs.append((e, slo, sln, D.slab_layout(slo, sln)))
        stride = max(1, total // 65536)
        want = full[::stride].clone()

        def ostep():
            works = []
            for e, slo, sln, lay in subs:
                dst = full[slo:slo + sln]
                if args.callback == "jacobian":
                    e.eval_jacobian_dev(Z.data_ptr(), dst.data_ptr(), stream)
                else:
                    e.eval_hessian_dev(Z.data_ptr(), 1.0, mu.data_ptr(), dst.data_ptr(), stream)
                works += D.gather_slabs_async(full, lay)
            for w in works:
                w.wait()
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
        for e, *_ in subs:
            e.close()
    except Exception as e:  # reported, never fatal
        over_err = repr(e)
    tt = torch.tensor([t_both, t_gather, t_over], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_both, t_gather, t_over = (float(x) for x in tt.tolist())
    chk = bool(torch.isfinite(full[::max(1, total // 4096)]).all().item())
    res = {"n_ranks": dist.get_world_size(), "backend": args.backend, "bytes_per_rank_vector": 8.0 * total,
           "collective": ("one in-place all_gather_into_tensor on the padded vector" if gbuf is not None and (args.backend == "nccl" or not full.is_cuda)
                          else "one in-place broadcast per rank"),
           "ms_per_step_compute_and_gather": t_both / args.steps * 1e3, "gather_ms": t_gather / args.steps * 1e3,
           "gather_gbs_per_rank_received": 8.0 * (total - ln) / (t_gather / args.steps) / 1e9,
           "knot_points_per_s_with_gather": N_total * args.steps / t_both, "sampled_finite": chk,
           "overlap": "ms_per_step_compute_and_gather: none, the gather follows the callback; ms_per_step_overlapped: each rank's "
                      "knots over two handles, the first half's slabs travel while the second half computes (DESIGN.md section 6)"}
    if t_over == t_over:
        res.update({"ms_per_step_overlapped": t_over / args.steps * 1e3, "knot_points_per_s_overlapped": N_total * args.steps / t_over,
                    "overlapped_vs_sequential_max_rel_diff_sampled": over_diff})
    if over_err:
        res["overlapped_error"] = over_err
    return res, mine


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
    args = ap.parse_args()

    import numpy as np
    import torch
    import dto_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    N_total = Nk * world
    prob = dto_amd.host.synthetic.make_scaled_problem(N_total, n, m, seed=42)
    k_lo, k_hi = rank * Nk + 1, (rank + 1) * Nk
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

    def collect():
        """HIP-event figures of the engine's kernels since the last profile_reset."""
        ms_g, n_g, fl_g = ev.profile_get("bgemm")
        ms_s, n_s, fl_s = ev.profile_get("expmv")
        var = {}
        for key, nm in (("horner", "bgemm_horner"), ("square", "bgemm_square"), ("plain", "bgemm_plain"), ("basis", "basis")):
            ms_v, n_v, fl_v = ev.profile_get(nm)
            if n_v:
                var[key] = {"launches": n_v, "avg_launch_ms": ms_v / n_v, "tflops": fl_v / (ms_v * 1e-3) / 1e12}
                # the polynomial products ("horner") also sit on the HBM roof: per launch two operands, three or four power
                # matrices in the epilogue and one or two outputs -- seven npad x npad matrices per interval in each of the
                # three launches of the order-26 form; a squaring moves two
                streams = {"horner": 7, "square": 2}.get(key)
                if streams and (n, Nk) == (256, 2000):
                    gbs = streams * 8.0 * n * n * (Nk - 1) / (ms_v / n_v * 1e-3) / 1e9
                    var[key].update({"algorithmic_hbm_bytes": streams * 8.0 * n * n * (Nk - 1), "hbm_gbs": gbs,
                                     "hbm_frac": gbs / HBM_PEAK_GBS})
        return ms_g, n_g, fl_g, ms_s, n_s, fl_s, var

    ms_gemm, n_gemm, fl_gemm, ms_sweep, n_sweep, fl_sweep, variants = collect()
    # In the timed region the Jacobian's generator sweep shares the chip with the chain's products (option overlap_sweep, on
    # by default): a kernel's HIP-event duration there includes what it waited for the other stream.  The roofline of the
    # dominant kernel is therefore taken from a SERIAL pass -- the same K steps again with overlap_sweep = 0, one kernel at
    # a time, which is also what `rocprofv3 -- python3 bench.py --serial-kernels` shows (profiles/) -- and the timed
    # region's own figures are reported next to it.
    overlapped = None
    if args.callback == "jacobian" and not args.serial_kernels and not args.no_kernel_timing:
        overlapped = {"avg_launch_ms": ms_gemm / max(n_gemm, 1), "launches": n_gemm,
                      "achieved": fl_gemm / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0,
                      "sweep_ms_per_step": ms_sweep / args.steps,
                      "note": "k_bgemm launches of the timed region: the sweep runs next to them on a second stream"}
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
        ms_gemm, n_gemm, fl_gemm, ms_sweep, n_sweep, fl_sweep, variants = collect()
        overlapped["ms_per_step_serial_pass"] = serial_elapsed / args.steps * 1e3
    smax, terms = ev.last_stats()
    finite = bool(torch.isfinite(out).all().item())

    gather = None
    want_gather = args.gather if args.gather is not None else world > 1
    if want_gather and dist is not None and args.callback in ("jacobian", "hessian"):
        n_out = out.numel()
        del out  # its place is taken by the rank's slice of the full vector
        try:
            gather, out = measure_gather(args, dto_amd, torch, dist, ev, prob, dev, Z, stream, fence, N_total,
                                         mu if args.callback == "hessian" else None)
        except Exception as e:  # the gather is a report next to `value`, never a reason to lose the line
            gather, out = {"error": repr(e)}, torch.empty(n_out, dtype=torch.float64, device=dev)

    traffic, traffic_src = None, None
    try:  # HBM bytes per launch of the dominant kernel come from committed PMC passes (bench.py cannot run rocprofv3 on itself)
        tj = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic_k_bgemm.json")))
        if (n, Nk, args.callback) == (256, 2000, "jacobian"):
            traffic, traffic_src = tj["avg_per_launch_bytes"], tj["source"]
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
            "metric": baseline_metric() if (args.callback, n, Nk) == ("jacobian", 256, 2000)
            else f"knot-points/sec for eval_{args.callback}, {n}-state x {Nk}-knot bilinear",
            "value": N_total * args.steps / elapsed,
            "unit": "knot-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"configs[2]: {n}-state bilinear, {m} drives, N={Nk} knots per GPU "
                                   f"(make_scaled_problem shape, Philox seed 42), callback={args.callback}",
                       "state_dim": n, "drives": m, "knots_per_gpu": Nk, "knots_total": N_total,
                       "parallelism": f"knot-range shards x{world}", "outputs_finite": finite,
                       "max_squarings": smax, "sweep_terms": terms},
            "roofline": {
                "bound": "mfma", "kernel": "k_bgemm (batched FP64 MFMA GEMM of the propagator chain)",
                "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
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
        if gather is not None:
            line["gather"] = gather
        if world == 1 and args.callback == "jacobian" and not args.no_other_callbacks:
            # the other callbacks of the same problem, same protocol (3 untimed + 5 timed calls each): reported for
            # context (SURVEY.md §8d lists them next to the headline), never part of `value`
            line["other_callbacks"] = other_callbacks(dto_amd, torch, prob, ev, dev, Z, stream, N_total)
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
