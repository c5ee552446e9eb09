#!/bin/bash
# A/B of the tuning switches of the TUNING=1 build (libdto_engine_t.so) inside ONE gpurun call: tools/ab_env_t.sh "K1=V1 K2=V2" ["..."] ...
# each configuration (plus the default) runs REPS times (default 3), interleaved.
# columns: ms/step, polynomial product, squaring (ms per launch), sweep (ms per step), generator-subspace GEMM (ms per launch), serial-pass ms/step, finite
export DTO_ENGINE_LIB=libdto_engine_t.so
run() { env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-callbacks 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['template_instances']; print('$*', round(d['ms_per_step'],3), round(r['horner']['avg_launch_ms'],4), round(r['square']['avg_launch_ms'],4), round(d['secondary_kernel']['ms_per_step'],3), round(r['basis']['avg_launch_ms'],4), round(d['roofline']['timed_region']['ms_per_step_serial_pass'],3), d['config']['outputs_finite'])"; }
for rep in $(seq 1 ${REPS:-3}); do
  run DEFAULT=1
  for cfg in "$@"; do run $cfg; done
done
