#!/usr/bin/env python3
"""Timeline of ONE callback out of tools/trace_call.sh's trace: kernels (K), copies (M) and HIP API calls in start order with
durations, from the start of the n-th-from-last k_norm_bounds launch to the next one.  usage: trace_view.py <trace dir> [n=4] [min_us=0]"""
import csv
import glob
import sys

d = sys.argv[1]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
K = list(csv.DictReader(open(glob.glob(d + "/*/*_kernel_trace.csv")[0])))
A = list(csv.DictReader(open(glob.glob(d + "/*/*_hip_api_trace.csv")[0])))
Mf = glob.glob(d + "/*/*_memory_copy_trace.csv")
M = list(csv.DictReader(open(Mf[0]))) if Mf else []
marks = sorted(int(k["Start_Timestamp"]) for k in K if "k_norm_bounds" in k["Kernel_Name"])
t0, t1 = marks[-nth] - 60000, marks[-nth + 1] - 60000
ev = []
for k in K:
    s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
    if t0 <= s < t1:
        ev.append((s, e, "K q%s %s" % (k["Queue_Id"], k["Kernel_Name"][:64])))
for a in A:
    s, e = int(a["Start_Timestamp"]), int(a["End_Timestamp"])
    if t0 <= s < t1 and (e - s) / 1e3 >= min_us:
        ev.append((s, e, "    api " + a["Function"]))
for m in M:
    s, e = int(m["Start_Timestamp"]), int(m["End_Timestamp"])
    if t0 <= s < t1:
        ev.append((s, e, "M " + m["Direction"]))
ev.sort()
for s, e, n in ev:
    print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:8.1f} {n}")
