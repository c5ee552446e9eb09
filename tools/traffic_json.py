#!/usr/bin/env python3
"""profiles/<tag>_traffic_jacobian_256x2000.json from the two PMC passes of tools/profile_round.sh (a tools/ helper):
  tools/traffic_json.py <tag> <pmc_fetch counter_collection.csv> <pmc_write counter_collection.csv> <calls in each pass>
HBM-side bytes per launch and kernel (FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for 16-byte-per-lane
streaming reads; WRITE_SIZE as read; both counters are in KB), the k_bgemm family's average (what bench.py's
`roofline.traffic` quotes) and the total per eval_constraint_jacobian call."""
import collections
import csv
import json
import re
import sys


def main():
    tag, f_fetch, f_write, calls = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    agg = collections.OrderedDict()
    for idx, f in ((0, f_fetch), (1, f_write)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not (k.startswith(("void dto", "dto::")) or "fillBuffer" in k):
                continue
            if "k_add" in k or "GemmShape<64, 64" in k or "k_norm1" in k:
                continue  # create-time work (generator-subspace products, generator norms): not part of a callback
            a = agg.setdefault(k, [0, 0.0, 0, 0.0])
            a[2 * idx] += 1
            a[2 * idx + 1] += float(r["Counter_Value"]) * 1024.0
    kernels, total = {}, 0.0
    for k, a in agg.items():
        n = max(a[0], a[2], 1)
        read, write = 2.0 * a[1] / max(a[0], 1), a[3] / max(a[2], 1)
        per_call = (2.0 * a[1] + a[3] * (a[0] / max(a[2], 1) if a[2] else 0)) / calls if a[0] else a[3] / calls
        kernels[k[:120]] = {"launches_per_call": n / calls, "read": read, "write": write, "per_call": (read + write) * n / calls}
        total += (read + write) * n / calls
    bg = {k: v for k, v in kernels.items() if "k_bgemm" in k and v["read"] + v["write"] > 1e9}
    prods = {k: v for k, v in bg.items() if ", 2, " not in k.split(">")[1] if True}
    out = {"kernel": "k_bgemm", "workload": "256-state x 2000-knot, eval_constraint_jacobian (three-product form: 3 polynomial products + 1 squaring per call)",
           "source": f"profiles/{tag}_bench_jacobian_256x2000.md (separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the round's final "
                     f"build, {calls} calls each; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section)",
           "per_kernel_bytes": kernels,
           "per_launch_bytes": {}, "per_call_total_bytes": total,
           "algorithmic_per_call_bytes": 8.0 * (275462200 + 530000 + 5 * 256 * 256)}
    i = 0
    for k, v in bg.items():
        m = re.search(r"k_bgemm_r<(\d+),", k) or re.search(r">, (\d+),", k)   # epilogue kind: ring kernel / double-buffered kernels
        epi = m.group(1) if m else "?"
        name = "square" if epi == "2" else f"product{ {'3': 1, '4': 2, '1': 3}.get(epi, 9) }"
        out["per_launch_bytes"][name] = {"read": v["read"], "write": v["write"], "kernel": k[:100]}
    pl = out["per_launch_bytes"]
    out["avg_per_launch_bytes"] = sum(v["read"] + v["write"] for v in pl.values()) / max(len(pl), 1)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
