#!/bin/bash
# A/B of the Hessian's fill!(H, 0) on the second stream (DTO_HESS_ZERO_BESIDE, TUNING build; run through gpurun from the repo root)
mkdir -p gpurun_out/r04w
: > gpurun_out/r04w/ab_hess.log
for shape in "256 2000" "64 1000" "256 250"; do set -- $shape; for zb in 1 0 1 0; do
DTO_ENGINE_LIB=libdto_engine_t.so DTO_HESS_ZERO_BESIDE=$zb timeout -k 10 300 python bench.py --states $1 --knots $2 --callback hessian --steps 10 --warmup 3 --no-cpu-baseline --no-other-callbacks --no-bound-output 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 x $2 zero_beside=$zb', round(d['ms_per_step'],3), d['config']['outputs_finite'])" >> gpurun_out/r04w/ab_hess.log
done; done
cat gpurun_out/r04w/ab_hess.log
