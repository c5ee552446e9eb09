#!/usr/bin/env python3
"""Vendor yardstick for the propagator chain's batched FP64 GEMM (a tools/ probe, never on the product path):
torch.bmm on B x n x n x n doubles (rocBLAS / hipBLASLt behind it), timed with HIP events on torch's stream.

  python3 tools/bmm_yardstick.py [B=2000] [n=256] [reps=20]
  rocprofv3 --kernel-trace --stats -- python3 tools/bmm_yardstick.py   # names the vendor kernel and its duration

Prints one JSON line: TFLOP/s of C = A B (three distinct matrices per batch entry, streamed from HBM) and of the
squaring C = A A, to set beside tools/bgemm_probe2 (the engine's own GEMM template on the same shape).

A fourth argument `sustain` (seconds) keeps the product running back to back for that long and reports the rate of every 0.1 s
window: the chip lowers its clock under a sustained FP64 matrix load (MI355X_MICROARCH.md, DVFS give-back), so the rate of
the first 20 launches is NOT the rate a 12 ms callback full of such products sees."""
import json
import sys

import torch


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(B, n, n, dtype=torch.float64, device=dev, generator=g)
    Bm = torch.randn(B, n, n, dtype=torch.float64, device=dev, generator=g)
    C = torch.empty_like(A)
    out = {"batch": B, "n": n, "reps": reps, "flops_per_call": 2.0 * n ** 3 * B}
    for name, fn in (("product", lambda: torch.bmm(A, Bm, out=C)), ("square", lambda: torch.bmm(A, A, out=C))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": ms, "tflops": 2.0 * n ** 3 * B / (ms * 1e-3) / 1e12}
    sustain = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    if sustain > 0:
        for name, fn in (("product", lambda: torch.bmm(A, Bm, out=C)), ("square", lambda: torch.bmm(A, A, out=C))):
            import time
            torch.cuda.synchronize()
            t_end = time.perf_counter() + sustain
            windows = []
            while time.perf_counter() < t_end:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = max(1, int(0.1 / (out[name]["ms"] * 1e-3)))
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                windows.append(2.0 * (A.shape[1] ** 3) * B * n / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            out[name]["sustained_tflops_windows"] = [round(w, 1) for w in windows]
            out[name]["sustained_tflops_last"] = windows[-1]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
