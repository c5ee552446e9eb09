#!/bin/bash
# A/B of engine BUILDS inside one gpurun call: tools/ab_lib.sh libdto_engine_x.so [...]; default build interleaved, 3 repetitions
run() { DTO_ENGINE_LIB=$1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-callbacks 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['template_instances']; print('$1', round(d['ms_per_step'],3), round(r['horner']['avg_launch_ms'],4), round(r['square']['avg_launch_ms'],4), round(d['secondary_kernel']['ms_per_step'],3), d['config']['outputs_finite'])"; }
for rep in 1 2 3; do
  run libdto_engine.so
  for lib in "$@"; do run $lib; done
done
