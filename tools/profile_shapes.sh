#!/bin/bash
# Per-shape measurement artifacts of the BASELINE configurations other than the headline's (run through gpurun from the repo root):
#   tools/profile_shapes.sh <tag>
# for every (states x knots, callback): rocprofv3 kernel-trace stats of `bench.py --serial-kernels` (one kernel at a time) and the
# bench line itself (default run).  Lands under gpurun_out/shapes_<tag>/; tools/summarize_shapes.py turns it into profiles/<tag>_*.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/shapes_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "64 1000 jacobian" "64 1000 hessian" "64 1000 constraint" "256 250 jacobian" "256 500 jacobian" "256 2000 constraint" \
            "1024 500 jacobian" "1024 500 hessian"; do
  set -- $spec
  n=$1; k=$2; cb=$3
  name=${n}x${k}_${cb}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py --states $n --knots $k --callback $cb \
      --steps 5 --warmup 2 --no-cpu-baseline --no-other-callbacks --no-bound-output --serial-kernels > $O/prof_$name.log 2>&1 || echo "rocprof $name failed"
  (cd $R && timeout -k 10 300 python bench.py --states $n --knots $k --callback $cb --no-cpu-baseline --no-other-callbacks > $O/bench_$name.log 2>&1) || echo "bench $name failed"
  echo "$name: $(tail -1 $O/bench_$name.log | cut -c1-200)"
done
