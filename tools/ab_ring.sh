#!/bin/bash
# ring core against the double-buffered core for the chain's batched GEMM (TUNING build, one gpurun call): tools/ab_ring.sh
# columns: ms/step, per-launch ms of the polynomial products / squaring / generator-subspace GEMMs (serial pass), serial-pass ms/step
export DTO_ENGINE_LIB=libdto_engine_t.so
run() { env "${@:3}" timeout -k 10 300 python bench.py --n $1 --knots $2 --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-other-callbacks --no-bound-output 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['template_instances']; print('$*', round(d['ms_per_step'],3), {k:round(v['avg_launch_ms'],4) for k,v in r.items()}, round(d['roofline']['timed_region']['ms_per_step_serial_pass'],3), d['config']['outputs_finite'])"; }
for rep in 1 2; do
  run 256 2000 DTO_BGEMM_RING=0; run 256 2000 DTO_BGEMM_RING=-1; run 256 2000 DTO_BGEMM_RING=1
done
STEPS=3
for rep in 1 2; do
  run 1024 500 DTO_BGEMM_RING=0; run 1024 500 DTO_BGEMM_RING=-1
  run 512 500 DTO_BGEMM_RING=0; run 512 500 DTO_BGEMM_RING=-1
done
