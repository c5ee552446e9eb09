#!/bin/bash
# A/B of runtime tuning knobs inside ONE gpurun call (cross-call timings drift by a few %):  tools/ab_env.sh "K1=V1 K2=V2" ["..."] ...
# each configuration (plus the default) is run three times, interleaved.
# columns: ms/step, polynomial product, squaring (ms per launch), sweep (ms per step), generator-subspace GEMM (ms per launch), finite
run() { env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-callbacks 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['template_instances']; print('$*', round(d['ms_per_step'],3), round(r['horner']['avg_launch_ms'],4), round(r['square']['avg_launch_ms'],4), round(d['secondary_kernel']['ms_per_step'],3), round(r['basis']['avg_launch_ms'],4), d['config']['outputs_finite'])"; }
for rep in 1 2 3; do
  run DEFAULT=1
  for cfg in "$@"; do run $cfg; done
done
