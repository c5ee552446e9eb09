#!/bin/bash
# Which unit binds the polynomial products?  One rocprofv3 --pmc pass per counter group (separate passes, --kernel-trace only,
# as MI355X_MICROARCH.md prescribes) over `bench.py --serial-kernels` (one kernel at a time), then a clock / power trace of a
# 200-step run (tools/clock_power_trace.py).  Run through gpurun from the repo root:  tools/profile_products.sh <tag>
# Lands under gpurun_out/products_<tag>/; tools/summarize_products.py turns it into profiles/<tag>_products_*.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/products_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-other-callbacks --no-bound-output --serial-kernels --steps 2 --warmup 1"
pass() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- $B > $O/pmc_$name.log 2>&1 || echo "pass $name failed: $(tail -2 $O/pmc_$name.log)"
}
pass sq_busy   SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass sq_insts  SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass tcc_ea    TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE GRBM_EA_BUSY
pass tcc_stall TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_REQ_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE GRBM_TC_BUSY
pass tcc_credit TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass tcc_level TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_SRC_FIFO_FULL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE
pass tcp       TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
cd $R
timeout -k 10 240 python tools/clock_power_trace.py $O/clock_power.csv > $O/clock_power.log 2>&1 || echo "clock/power trace failed"
tail -3 $O/clock_power.log
