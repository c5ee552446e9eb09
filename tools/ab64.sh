#!/bin/bash
# A/B of the 64-state forms on the 64 x 1000 callbacks (TUNING build; run through gpurun from the repo root):
#   DTO_CHAIN64 = 0/1  one-launch propagator chain,  DTO_SWEEP_S64 = 0/1  generator-stationary sweep,
#   DTO_SWEEP_EARLY = 0/1  Jacobian sweep planned from the cheap bound and enqueued before the chain
mkdir -p gpurun_out/r04w
: > gpurun_out/r04w/ab64.log
for cfg in ${AB64_CFGS:-"1 1 1" "1 1 0" "1 0 0" "0 0 0"}; do set -- $cfg; for cb in ${AB64_CBS:-jacobian hessian constraint}; do
DTO_ENGINE_LIB=libdto_engine_t.so DTO_CHAIN64=$1 DTO_SWEEP_S64=$2 DTO_SWEEP_EARLY=$3 timeout -k 10 200 python bench.py --states 64 --knots 1000 --callback $cb --no-cpu-baseline --no-other-callbacks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain64=$1 s64=$2 early=$3', d['metric'][:40], round(d['ms_per_step'],3), d['config']['outputs_finite'], d['config']['sweep_terms'], 'sweep', round(d.get('secondary_kernel',{}).get('ms_per_step',0),3), 'chain', round(d['roofline']['avg_launch_ms'],3), 'serial', round((d['roofline'].get('timed_region') or {}).get('ms_per_step_serial_pass',0),3))" >> gpurun_out/r04w/ab64.log
done; done
cat gpurun_out/r04w/ab64.log
