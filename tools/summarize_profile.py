#!/usr/bin/env python3
"""Turn rocprofv3 outputs (gpurun_out/...) into the committed summaries under profiles/.
usage: summarize_profile.py <round tag> <kernel_stats.csv> [<pmc_fetch counter_collection.csv> <pmc_write ...> [<pmc_sq ...>]]"""
import collections
import csv
import sys


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = [f"# rocprofv3 summary {tag}", "", "command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-callbacks --serial-kernels`",
           "(2 warm-up + 5 timed + 5 individually timed eval_constraint_jacobian calls, 256-state x 2000-knot bilinear, 1x MI355X; --serial-kernels = "
           "option overlap_sweep 0, one kernel at a time; the 64x64 k_bgemm / k_add launches are create-time work)", "",
           "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in list(csv.DictReader(open(stats)))[:16]:
        out.append("| `%s` | %s | %.3f | %.1f | %s |" % (r["Name"][:80].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       float(r["AverageNs"]) / 1e3, r["Percentage"]))
    if len(sys.argv) >= 5:
        out += ["", "## HBM traffic per launch (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, 5 calls each)", "",
                "FETCH_SIZE is doubled as MI355X_MICROARCH.md §HBM prescribes for 16-B-per-lane streaming reads "
                "(the counter tallies 128-B requests at 64 B); WRITE_SIZE is taken as read.", "",
                "| kernel | launches | raw FETCH_SIZE MB | corrected read MB | WRITE_SIZE MB |", "|---|---|---|---|---|"]
        agg = {}
        for idx, f in ((0, sys.argv[3]), (1, sys.argv[4])):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"][:70]
                a = agg.setdefault(k, [0, 0.0, 0, 0.0])
                a[2 * idx] += 1
                a[2 * idx + 1] += float(r["Counter_Value"])
        for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:10]:
            if not k.startswith(("void dto", "dto::", "__amd_rocclr_fill")):
                continue
            f = a[1] / max(a[0], 1) / 1024.0
            w = a[3] / max(a[2], 1) / 1024.0
            out.append("| `%s` | %d | %.1f | %.1f | %.1f |" % (k.replace("|", "/"), a[0], f, 2 * f, w))
    if len(sys.argv) >= 6:
        names = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                 "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
        out += ["", "## SQ counters (one `--pmc` pass: " + " ".join(names) + ")", "",
                "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8); clock = GRBM_GUI_ACTIVE / 8 / duration.", "",
                "| kernel | launches | avg us | clock GHz | MFMA util | wait_inst/wave | wait_any/wave | LDS conflict/active |",
                "|---|---|---|---|---|---|---|---|"]
        disp = {}
        for r in csv.DictReader(open(sys.argv[5])):
            d = disp.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"][:80], "t": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
        agg = collections.OrderedDict()
        for d in disp.values():
            if not d["k"].startswith(("void dto", "dto::")):
                continue
            a = agg.setdefault(d["k"], collections.Counter())
            a["n"] += 1
            a["t"] += d["t"]
            for nm in names:
                a[nm] += d.get(nm, 0.0)
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"])[:5]:
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0
            out.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
                k.replace("|", "/"), a["n"], a["t"] / a["n"] / 1e3, cyc / a["t"], a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
                a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1), a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1),
                a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)))
    print("\n".join(out))


if __name__ == "__main__":
    main()
