#!/bin/bash
# 64 x 64 against 128 x 128 tiles of the chain's batched GEMM at short shards (TUNING build, one gpurun call): tools/ab_tile64.sh
# columns: ms/step, per-launch ms of the polynomial products / squaring / generator-subspace GEMMs (serial pass), serial-pass ms/step
export DTO_ENGINE_LIB=libdto_engine_t.so
run() { env "$@" timeout -k 10 120 python bench.py --knots $K --steps 20 --warmup 3 --no-cpu-baseline --no-other-callbacks --no-bound-output 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['template_instances']; print('$* K=$K', round(d['ms_per_step'],3), {k:round(v['avg_launch_ms'],4) for k,v in r.items()}, round(d['roofline']['timed_region']['ms_per_step_serial_pass'],3))"; }
for K in 250 500 125; do for rep in 1 2; do run DEFAULT=1; run DTO_BGEMM_TILE64=0; done; done
