"""GPU: `bench.py` honours the driver's contract on a small shape -- one JSON line on stdout with the required keys, the
roofline block (serial pass) and the CPU baseline block; and the two-rank one-device rehearsal of the multi-GPU path prints
the gather block (sequential and overlapped forms)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--n", "128", "--knots", "1200",
                        "--cpu-budget", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["value"] > 0
    assert abs(d["value"] - 1200 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1 and rf["launches"] > 0
    assert rf["timed_region"]["launches"] == rf["launches"]          # the serial pass repeats the timed region's launches
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert d["config"]["outputs_finite"] is True and "HBM" in d["config"]["inputs"]
    # median of individually fenced steps next to the mean `value` is; per-call traffic key; the Hessian's own roofline
    assert d["ms_per_step_median"] > 0 and abs(d["value_from_median"] - 1200 / (d["ms_per_step_median"] * 1e-3)) <= 1e-6 * d["value"]
    assert "traffic_per_call" in rf
    hr = d["other_callbacks"]["eval_hessian_lagrangian"]
    assert hr["callback_hbm"]["algorithmic_bytes"] > 0
    si = d["other_callbacks"]["solver_iteration"]   # g, grad f, J, H at a new point per iteration, without / with the shared forward sweep
    assert si["finite"] is True and si["ms_per_iteration"] > 0 and si["ms_per_iteration_reuse_forward_sweep"] > 0, si
    if "roofline" in hr:   # (128 states x 1200 knots: the adjoint sweep runs in the fused or the cluster form, both timed)
        assert hr["roofline"]["bound"] == "mfma" and 0 < hr["roofline"]["frac"] < 1
    # round 4: the bandwidth-bound assembly kernels against the HBM roof, the PCIe-inclusive figures and SURVEY 8d's literal metric
    ah = d["assembly_hbm"]
    for k in ("zero_fill", "build_A", "basis_multi"):
        assert ah[k]["unit"] == "GB/s" and ah[k]["achieved"] > 0 and 0 < ah[k]["frac"] < 1.2 and ah[k]["algorithmic_bytes_per_launch"] > 0, (k, ah[k])
    hp = d["host_pointer"]
    assert hp["eval_constraint_jacobian"]["ms_per_call_median"] > 0 and hp["eval_hessian_lagrangian"]["ms_per_call_median"] > 0, hp
    vh = d["value_incl_h2d_median"]
    assert vh["value"] > 0 and vh["ms_per_step_median"] >= 0.9 * d["ms_per_step_median"]
    # the per-instance HBM figures exist for every shape, not only the headline's
    assert rf["template_instances"]["horner"]["algorithmic_hbm_bytes"] == 7 * 8.0 * 128 * 128 * 1199


def test_64_states_report_the_one_launch_chain():
    """configs[1]'s shape: the propagator chain is k_chain64 (one launch per call), the sweep the generator-stationary form; the line says so."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--n", "64", "--knots", "1000",
                        "--no-cpu-baseline", "--no-other-callbacks"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    rf = d["roofline"]
    assert "k_chain64" in rf["kernel"] and rf["launches"] == 3 and 0 < rf["frac"] < 1, rf
    assert set(rf["template_instances"]) >= {"chain64"} and "horner" not in rf["template_instances"]
    assert d["config"]["outputs_finite"] is True and d["assembly_hbm"]["zero_fill"]["achieved"] > 0


@pytest.mark.parametrize("callback", ["hessian", "constraint"])
def test_other_callbacks_have_a_roofline_line_of_their_own(callback):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--n", "256", "--knots", "300",
                        "--callback", callback, "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    rf = d["roofline"]
    assert callback in d["metric"] and rf["bound"] == "mfma" and rf["launches"] > 0 and 0 < rf["frac"] < 1, rf
    assert d["callback_hbm"]["algorithmic_bytes"] > 0 and d["config"]["outputs_finite"] is True


def test_four_rank_rehearsal_on_one_device():
    """The widest multi-rank rehearsal a one-GPU box allows (its process guard admits six processes with the GPU open: this test's
    own process, the launcher and four ranks -- five ranks were killed by it): four ranks, each with an engine handle on its shard
    plus the two sub-handles of the overlapped gather, four RCCL communicators over loopback, the in-place all-gather with four
    chunks and the strong split -- what `--gpus 8` runs on a node, at half the width.  World 8 itself is covered on the CPU: tests/test_comm_layout.py (layouts of configs[3] / configs[4] over
    eight ranks) and tests/test_distributed_gloo.py (eight gloo processes)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--one-device",
                        "--n", "64", "--knots", "200", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 4 and d["config"]["knots_total"] == 800 and "status" not in d
    g = d["gather"]
    assert "error" not in g and "overlapped_error" not in g, g
    assert g["n_ranks"] == 4 and g["sampled_finite"] is True and "ncclAllGather" in g["collective"]
    assert g["overlapped_vs_sequential_max_rel_diff_sampled"] <= 1e-10
    st = d["strong_scaling"]
    assert st["knots_total"] == 200 and st["knots_per_gpu"] == 50 and st["outputs_finite"]


def test_gpus_2_without_torchrun_starts_two_ranks_itself():
    """`bench.py --gpus 2` started plainly (WORLD_SIZE unset) launches its two ranks as a child torchrun before touching the
    GPU and relays ONE line with n_gpus = 2; the gather block goes through the engine's C-ABI collectives (two real RCCL ranks:
    on one device they pose as two hosts and use the socket transport), the strong-scaling block splits the same knots."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--n", "64", "--knots", "600", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["knots_total"] == 1200 and d["scaling"] == "weak"
    g = d["gather"]
    assert "error" not in g and "overlapped_error" not in g, g
    assert g["n_ranks"] == 2 and g["sampled_finite"] is True and "ncclAllGather" in g["collective"] and "C ABI" in g["transport"]
    assert g["ms_per_step_overlapped"] > 0 and g["overlapped_vs_sequential_max_rel_diff_sampled"] <= 1e-10
    st = d["strong_scaling"]
    assert st["scaling"] == "strong" and st["knots_total"] == 600 and st["knots_per_gpu"] == 300 and st["value"] > 0 and st["outputs_finite"]


def test_strong_scaling_line_under_torchrun():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--states", "64", "--knots", "600", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--no-gather"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["knots_total"] == 600 and d["config"]["knots_per_gpu"] == 300
    assert abs(d["value"] - 600 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_asking_for_more_gpus_than_visible_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "7", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)


def test_a_hung_report_block_cannot_lose_the_headline():
    """The gather / strong-scaling reports run collectives after `value` has been measured; a watchdog bounds them.  With the
    limit at zero every rank leaves at once and rank 0 prints the headline alone, marked `"status": "collective_timeout"` -- and the
    process exits NON-ZERO: a hung collective must never read as a clean run to whoever launched the bench."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", DTO_BENCH_WATCHDOG_S="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--n", "64", "--knots", "600", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode != 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    for k in REQUIRED[:-1]:
        assert k in d, k
    assert d["n_gpus"] == 2 and "watchdog" in d and d["status"] == "collective_timeout" and d["value"] > 0
    assert abs(d["value"] - 1200 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
