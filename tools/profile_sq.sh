#!/bin/bash
# One SQ-counter pass + one kernel trace of the default bench (jacobian), for quick kernel iteration:
#   tools/profile_sq.sh <tag> [extra bench args]
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-other-callbacks $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- $B --steps 5 --warmup 2 > $O/prof_$tag.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d $O/pmc_sq_$tag -- $B --steps 2 --warmup 1 > $O/pmc_sq_$tag.log 2>&1
cd $R
python3 tools/summarize_profile.py $tag $(find $O/prof_$tag -name '*kernel_stats.csv' | head -1) /dev/null /dev/null $(find $O/pmc_sq_$tag -name '*counter_collection.csv' | head -1) > $O/summary_$tag.md
cat $O/summary_$tag.md
