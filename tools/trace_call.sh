#!/bin/bash
# Kernel + HIP API + copy trace of one callback shape in the default (overlapped) mode (run through gpurun from the repo root):
#   tools/trace_call.sh <states> <knots> <callback>   -> gpurun_out/trace_<states>x<knots>_<callback>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace_$1x$2_$3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/bench.py --states $1 --knots $2 --callback $3 --no-cpu-baseline --no-other-callbacks --no-bound-output --no-kernel-timing --steps 8 --warmup 3 > $O/bench.log 2>&1 < /dev/null
ls $O/*/ | head -3
