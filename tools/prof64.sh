#!/bin/bash
# Kernel trace of the 64-state x 1000-knot callbacks, one kernel at a time (run through gpurun from the repo root):
#   tools/prof64.sh [callbacks...]   -> gpurun_out/r04w/p64/<callback>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04w/p64
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cb in ${@:-jacobian hessian constraint}; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$cb -- python3 $R/bench.py --states 64 --knots 1000 --callback $cb --no-cpu-baseline --no-other-callbacks --no-bound-output --serial-kernels --steps 20 --warmup 5 > $O/$cb.log 2>&1 < /dev/null
f=$(find $O/$cb -name '*kernel_stats.csv' | head -1)
echo "== $cb"; [ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-140 | head -14
done
