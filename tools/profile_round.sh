#!/bin/bash
# Collect the per-round measurement artifacts on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>
# kernel-trace stats, separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters) and the default bench line;
# everything lands under gpurun_out/, tools/summarize_profile.py turns it into profiles/<tag>_*.
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# per-kernel figures need one kernel at a time: --serial-kernels (option overlap_sweep = 0); the default run is traced once too
B="python3 $R/bench.py --no-cpu-baseline --no-other-callbacks --no-bound-output --serial-kernels"
BO="python3 $R/bench.py --no-cpu-baseline --no-other-callbacks --no-bound-output --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- $B --steps 5 --warmup 2 > $O/prof_$tag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_overlapped -- $BO --steps 5 --warmup 2 > $O/prof_${tag}_overlapped.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$tag -- $B --steps 2 --warmup 1 > $O/pmc_fetch_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$tag -- $B --steps 2 --warmup 1 > $O/pmc_write_$tag.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-trace --output-format csv -d $O/pmc_sq_$tag -- $B --steps 2 --warmup 1 > $O/pmc_sq_$tag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_hess -- $B --steps 5 --warmup 2 --callback hessian > $O/prof_${tag}_hess.log 2>&1
cd $R
python bench.py > $O/bench_$tag.log 2>&1
python bench.py --callback hessian --no-cpu-baseline > $O/bench_${tag}_hess.log 2>&1
python bench.py --callback constraint --no-cpu-baseline > $O/bench_${tag}_cons.log 2>&1
tail -1 $O/bench_$tag.log | cut -c1-600
