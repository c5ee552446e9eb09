#!/bin/bash
# SQ counters of the 64-state kernels (one rocprofv3 --pmc pass, kernel trace only; run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc64
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cb in jacobian hessian; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/$cb -- python3 $R/bench.py --states 64 --knots 1000 --callback $cb --no-cpu-baseline --no-other-callbacks --no-bound-output --serial-kernels --steps 3 --warmup 1 > $O/$cb.log 2>&1 < /dev/null
done
ls $O/*/*/ | head
