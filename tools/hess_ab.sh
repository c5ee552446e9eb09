#!/bin/bash
# A/B of the Hessian callback at 256 x 2000 with tuning switches (TUNING build): tools/hess_ab.sh "K=V" ...
export DTO_ENGINE_LIB=libdto_engine_t.so
run() { env "$@" timeout -k 10 120 python bench.py --callback hessian --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), d['config']['outputs_finite'])"; }
for rep in 1 2 3; do
  run DEFAULT=1
  for cfg in "$@"; do run $cfg; done
done
