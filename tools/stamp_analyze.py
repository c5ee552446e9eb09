#!/usr/bin/env python3
"""Phase stamps of the batched GEMM (TUNING build: DTO_STAMP_FILE / DTO_STAMP_LAUNCH, csrc/dto_kernels.hip) -> per-CU timeline summary.

  python3 tools/stamp_analyze.py gpurun_out/<dir>/stamps.*

Per tile the kernel records HW_ID | XCC_ID << 32, s_memtime at tile start / after the MFMA loop / after the epilogue's memory
operations have drained, and s_memrealtime (100 MHz) at start and end.  For every launch the script prints the duration of the
loop and epilogue phases and, per CU, how much of the launch had 0 / 1 / 2+ workgroups inside their MFMA loops -- i.e. whether the
co-resident workgroups alternate (one in its loop while the other streams its epilogue) or move in step."""
import sys

import numpy as np


def main():
    for path in sys.argv[1:]:
        raw = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
        raw = raw[raw[:, 3] != 0]
        if not len(raw):
            print(path, "no stamps")
            continue
        hw = raw[:, 0]
        cu = ((hw >> np.uint64(8)) & np.uint64(0xff)) | ((hw >> np.uint64(32)) << np.uint64(8))   # cu, sh, se | xcc
        t0, t1, t2 = (raw[:, i].astype(np.int64) for i in (1, 2, 3))
        r0, r1 = raw[:, 4].astype(np.int64), raw[:, 5].astype(np.int64)
        # s_memtime counters are per XCD (unsynchronised): durations come from s_memtime, positions from the 100 MHz real-time counter
        tick_ns = float(np.median(10.0 * (r1 - r0) / np.maximum(1, t2 - t0)))   # ns per s_memtime tick
        loop_us, epi_us = (t1 - t0) * tick_ns / 1e3, (t2 - t1) * tick_ns / 1e3
        vm_us, bar_us = raw[:, 6].astype(np.int64) * tick_ns / 1e3, raw[:, 7].astype(np.int64) * tick_ns / 1e3
        span_us = (r1.max() - r0.min()) * 10.0 / 1e3
        t1 = (r0 * 10.0 + (t1 - t0) * tick_ns) / tick_ns
        t0, t2 = r0 * 10.0 / tick_ns, np.maximum(r1 * 10.0 / tick_ns, t1)
        base = t0.min()
        print(f"{path}: {len(raw)} tiles on {len(np.unique(cu))} CUs, span {span_us:.1f} us, s_memtime tick {tick_ns:.2f} ns")
        for name, d in (("loop", loop_us), ("epilogue", epi_us), ("load wait", vm_us), ("barrier", bar_us)):
            print(f"  {name:9s} us: mean {d.mean():7.2f}  p10 {np.percentile(d, 10):7.2f}  p50 {np.percentile(d, 50):7.2f}  p90 {np.percentile(d, 90):7.2f}")
        # per CU: time with k workgroups in their loops / in their epilogues
        occ_loop = np.zeros(4)
        occ_epi = np.zeros(4)
        both = 0.0
        gaps = 0.0
        for c in np.unique(cu):
            sel = cu == c
            ev = []
            for a, b, e in zip(t0[sel], t1[sel], t2[sel]):
                ev += [(a, 0, 1), (b, 0, -1), (b, 1, 1), (e, 1, -1)]
            ev.sort()
            n = [0, 0]
            last = ev[0][0]
            for t, kind, d in ev:
                dt = (t - last) * tick_ns / 1e3
                occ_loop[min(n[0], 3)] += dt
                occ_epi[min(n[1], 3)] += dt
                if n[0] >= 1 and n[1] >= 1:
                    both += dt
                if n[0] == 0 and n[1] == 0:
                    gaps += dt
                n[kind] += d
                last = t
        tot = occ_loop.sum()
        print("  CU time with k workgroups in the MFMA loop : " + "  ".join(f"k={k}: {occ_loop[k] / tot:.3f}" for k in range(4)))
        print("  CU time with k workgroups in the epilogue  : " + "  ".join(f"k={k}: {occ_epi[k] / tot:.3f}" for k in range(4)))
        print(f"  loop beside an epilogue: {both / tot:.3f}   nothing resident: {gaps / tot:.3f}   mean busy span per CU {tot / len(np.unique(cu)):.1f} us")
        # one CU's first events, as a picture
        c = np.unique(cu)[len(np.unique(cu)) // 2]
        sel = np.where(cu == c)[0]
        sel = sel[np.argsort(t0[sel])][:8]
        print("  one CU, first tiles (start, loop end, end in us): " + "  ".join(
            f"[{(t0[i] - base) * tick_ns / 1e3:.1f} {(t1[i] - base) * tick_ns / 1e3:.1f} {(t2[i] - base) * tick_ns / 1e3:.1f}]" for i in sel))


if __name__ == "__main__":
    main()
