#!/usr/bin/env python3
"""gpurun_out/products_<tag>/ (tools/profile_products.sh) -> profiles/<tag>_products_counters.md (+ the clock / power trace):
per kernel of the propagator chain and per counter the mean per launch, and the derived ratios that say which unit is busy.
usage: summarize_products.py <tag>"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"products_{tag}")
KEEP = ("k_bgemm_r<1", "k_bgemm_r<2", "k_bgemm_r<3", "k_bgemm_r<4", "k_basis_gemm_multi", "k_basis_gemm<", "k_sweep_fused", "k_jac_zero", "k_build_A")


def short(name):
    for k in KEEP:
        if k in name:
            return name[name.index(k):][:40].split("(")[0]
    return None


data = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, n]
dur = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        a = data[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            d = dur[k]
            d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1


def mean(k, c):
    a = data[k].get(c)
    return a[0] / a[1] if a and a[1] else float("nan")


out = [f"# Which unit binds the polynomial products -- PMC passes {tag}", "",
       "`rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu-baseline --no-other-callbacks --no-bound-output --serial-kernels --steps 2 --warmup 1`,",
       "one pass per counter group (tools/profile_products.sh), 256-state x 2000-knot Jacobian, one kernel at a time.  Means per launch.",
       "`k_bgemm_r<1,4>` = first product (A^4 K -> Y+Pa, Y+Pb: two outputs, three epilogue streams), `<3,4>` = second (two outputs, four streams),",
       "`<4,4>` = third (one output + slab option), `<2,4>` = the squaring (no epilogue streams; stores -E_k into the Jacobian slab).", ""]
kern = [k for k in sorted(data) if dur[k][1]]
out += ["## Busy split (SQ counters; cycles are per-SIMD sums)", "",
        "| kernel | us | clock GHz | MFMA busy | MFMA+VALU co-exec / MFMA busy | VALU active / wave cycles | LDS active / wave cycles | VMEM active / wave cycles | wait_inst / wave | wait_any / wave |",
        "|---|---|---|---|---|---|---|---|---|---|"]
for k in kern:
    t = dur[k][0] / dur[k][1]
    cyc = mean(k, "GRBM_GUI_ACTIVE") / 8.0
    wc = mean(k, "SQ_WAVE_CYCLES")
    out.append("| `%s` | %.1f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
        k, t / 1e3, cyc / t, mean(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), mean(k, "SQ_VALU_MFMA_COEXEC_CYCLES") / max(mean(k, "SQ_VALU_MFMA_BUSY_CYCLES"), 1),
        mean(k, "SQ_ACTIVE_INST_VALU") / wc, mean(k, "SQ_ACTIVE_INST_LDS") / wc, mean(k, "SQ_ACTIVE_INST_VMEM") / wc,
        mean(k, "SQ_WAIT_INST_ANY") / wc, mean(k, "SQ_WAIT_ANY") / wc))
out += ["", "## Instruction mix per launch", "",
        "| kernel | MFMA F64 MOPS (x512 flop) | v_fma_f64 | v_mul_f64 | v_add_f64 | FP64 VALU per MFMA instr (MOPS/4) | LDS instr | VMEM rd | VMEM wr |", "|---|---|---|---|---|---|---|---|---|"]
for k in kern:
    mops = mean(k, "SQ_INSTS_VALU_MFMA_MOPS_F64")
    f64 = mean(k, "SQ_INSTS_VALU_FMA_F64") + mean(k, "SQ_INSTS_VALU_MUL_F64") + mean(k, "SQ_INSTS_VALU_ADD_F64")
    out.append("| `%s` | %.3g | %.3g | %.3g | %.3g | %.2f | %.3g | %.3g | %.3g |" % (
        k, mops, mean(k, "SQ_INSTS_VALU_FMA_F64"), mean(k, "SQ_INSTS_VALU_MUL_F64"), mean(k, "SQ_INSTS_VALU_ADD_F64"), f64 / max(mops / 4.0, 1),
        mean(k, "SQ_INSTS_LDS"), mean(k, "SQ_INSTS_VMEM_RD"), mean(k, "SQ_INSTS_VMEM_WR")))
out += ["", "## L2 <-> fabric (TCC counters, summed over the 128 channels)", "",
        "read bytes = 64 B x RDREQ (the 32-B requests at 32 B); write bytes = 64 B x WRREQ.  Stall ratios are per TCC cycle (TCC_CYCLE_sum = channels x cycles).", "",
        "| kernel | EA read GB | EA write GB | GB/s (r+w) | WRREQ_STALL / cycle | TAG_STALL / cycle | too-many-WRREQ stall / cycle | RD credit stall / cycle | WR credit stall / cycle | "
        "reads in flight per channel | writes in flight per channel | L2 hit rate | TCC busy / cycle | EA busy (GRBM) | TC busy (GRBM) |", "|" + "---|" * 15]
for k in kern:
    t = dur[k][0] / dur[k][1]
    rd, rd32, wr = mean(k, "TCC_EA0_RDREQ_sum"), mean(k, "TCC_EA0_RDREQ_32B_sum"), mean(k, "TCC_EA0_WRREQ_sum")
    rb = (rd - rd32) * 64.0 + rd32 * 32.0
    wb = wr * 64.0
    cyc = mean(k, "TCC_CYCLE_sum")
    g = mean(k, "GRBM_GUI_ACTIVE")
    out.append("| `%s` | %.2f | %.2f | %.0f | %.3f | %.3f | %.3f | %.3f | %.3f | %.1f | %.1f | %.2f | %.2f | %.2f | %.2f |" % (
        k, rb / 1e9, wb / 1e9, (rb + wb) / t, mean(k, "TCC_EA0_WRREQ_STALL_sum") / cyc, mean(k, "TCC_TAG_STALL_sum") / cyc,
        mean(k, "TCC_TOO_MANY_EA_WRREQS_STALL_sum") / cyc, mean(k, "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") / cyc, mean(k, "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum") / cyc,
        mean(k, "TCC_EA0_RDREQ_LEVEL_sum") / cyc, mean(k, "TCC_EA0_WRREQ_LEVEL_sum") / cyc,
        mean(k, "TCC_HIT_sum") / max(mean(k, "TCC_HIT_sum") + mean(k, "TCC_MISS_sum"), 1), mean(k, "TCC_BUSY_sum") / cyc,
        mean(k, "GRBM_EA_BUSY") / g, mean(k, "GRBM_TC_BUSY") / g))
out += ["", "## L1 (TCP counters, summed over the CUs)", "",
        "| kernel | pending-request stall cycles / (256 CU x cycles) | TCR->TCP stall / (256 x cycles) | mean L1->L2 read latency (cycles) | TA busy (GRBM) |", "|---|---|---|---|---|"]
for k in kern:
    cyc = mean(k, "GRBM_GUI_ACTIVE") / 8.0
    out.append("| `%s` | %.2f | %.2f | %.0f | %.2f |" % (
        k, mean(k, "TCP_PENDING_STALL_CYCLES_sum") / (256.0 * cyc), mean(k, "TCP_TCR_TCP_STALL_CYCLES_sum") / (256.0 * cyc),
        mean(k, "TCP_TCC_READ_REQ_LATENCY_sum") / max(mean(k, "TCP_TCC_READ_REQ_sum"), 1), mean(k, "GRBM_TA_BUSY") / mean(k, "GRBM_GUI_ACTIVE")))
cp = os.path.join(src, "clock_power.log")
if os.path.exists(cp):
    out += ["", "## Clock and board power over 200 back-to-back calls (tools/clock_power_trace.py: amdgpu hwmon sensors every 10 ms)", "", "```"] + \
           [l.rstrip()[:400] for l in open(cp) if l.startswith(("busy card", "bench line"))] + ["```"]
    shutil.copy(os.path.join(src, "clock_power.csv"), os.path.join(ROOT, "profiles", f"{tag}_clock_power_trace.csv"))
open(os.path.join(ROOT, "profiles", f"{tag}_products_counters.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
