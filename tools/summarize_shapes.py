#!/usr/bin/env python3
"""gpurun_out/shapes_<tag>/ (tools/profile_shapes.sh) -> profiles/<tag>_<shape>_<callback>_kernel_stats.csv, the bench lines, and one
markdown table per shape: usage: summarize_shapes.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"shapes_{tag}")
out = [f"# rocprofv3 summaries per shape, {tag}", "",
       "command per row group: `rocprofv3 --kernel-trace --stats -- python3 bench.py --states N --knots K --callback C --steps 5 --warmup 2 "
       "--no-cpu-baseline --no-other-callbacks --no-bound-output --serial-kernels` (7 calls + 5 individually fenced ones for the Jacobian; "
       "one kernel at a time), next to the line of the default run of the same shape (`bench_line`).", ""]
for log in sorted(glob.glob(os.path.join(src, "bench_*.log"))):
    name = os.path.basename(log)[len("bench_"):-len(".log")]
    try:
        line = json.loads([l for l in open(log).read().splitlines() if l.startswith("{")][-1])
    except Exception:
        out += [f"## {name}", "", "bench line missing", ""]
        continue
    json.dump(line, open(os.path.join(ROOT, "profiles", f"{tag}_bench_line_{name}.json"), "w"), indent=1)
    stats = glob.glob(os.path.join(src, f"prof_{name}", "*", "*_kernel_stats.csv"))
    rf = line["roofline"]
    out += [f"## {name}", "",
            f"bench line: **{line['ms_per_step']:.3f} ms per call**, {line['value']:.0f} knot-points/s; dominant kernel `{rf['kernel'][:60]}`: "
            f"{rf['achieved']:.1f} TFLOP/s = **{rf['frac']:.3f}** of {rf['peak']} ({rf['launches']} launches, {rf['avg_launch_ms']:.3f} ms each); "
            f"callback vs HBM roof: {line['callback_hbm']['algorithmic_bytes'] / 1e6:.1f} MB algorithmic -> {line['callback_hbm']['achieved']:.0f} GB/s "
            f"= {line['callback_hbm']['frac']:.4f}", ""]
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_{name}_kernel_stats.csv"))
        rows = list(csv.DictReader(open(stats[0])))
        calls = 12 if name.endswith("jacobian") else 7
        out += [f"| kernel | calls | total ms | avg us | % | ms per callback ({calls} calls traced) |", "|---|---|---|---|---|---|"]
        for r in rows[:12]:
            out.append("| `%s` | %s | %.3f | %.1f | %s | %.3f |" % (r["Name"][:70].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                 float(r["AverageNs"]) / 1e3, r["Percentage"], float(r["TotalDurationNs"]) / 1e6 / calls))
        tot = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith(("dto::", "void dto", "__amd_rocclr")))
        out += ["", f"sum of kernel time per callback: {tot / 1e6 / calls:.3f} ms", ""]
open(os.path.join(ROOT, "profiles", f"{tag}_shapes.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out)[:6000])
