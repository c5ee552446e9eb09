#!/bin/bash
# A/B of the 64-state forms on the 64 x 1000 callbacks (TUNING build; run through gpurun from the repo root):
#   DTO_CHAIN64 = 0/1  one-launch propagator chain,  DTO_SWEEP_S64 = 0/1  generator-stationary sweep
mkdir -p gpurun_out/r04w
: > gpurun_out/r04w/ab64.log
for cfg in "1 1" "1 0" "0 0"; do set -- $cfg; for cb in jacobian hessian constraint; do
DTO_ENGINE_LIB=libdto_engine_t.so DTO_CHAIN64=$1 DTO_SWEEP_S64=$2 timeout -k 10 200 python bench.py --states 64 --knots 1000 --callback $cb --no-cpu-baseline --no-other-callbacks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain64=$1 s64=$2', d['metric'][:40], round(d['ms_per_step'],3), d['config']['outputs_finite'], d['config']['sweep_terms'], d.get('secondary_kernel',{}).get('ms_per_step'))" >> gpurun_out/r04w/ab64.log
done; done
cat gpurun_out/r04w/ab64.log
