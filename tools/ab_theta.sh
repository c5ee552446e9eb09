#!/bin/bash
# The sub-stepping radius of the sweeps (DTO_THETA_V, TUNING build): Hessian time and error against the oracle where the default 9 forces
# q = 2 rounds (which takes the Hessian off the pairing path).  Run through gpurun from the repo root.
mkdir -p gpurun_out/r04w
: > gpurun_out/r04w/ab_theta.log
for th in 9 12 15 18; do
DTO_ENGINE_LIB=libdto_engine_t.so DTO_THETA_V=$th timeout -k 10 300 python bench.py --states 1024 --knots 500 --callback hessian --steps 3 --warmup 1 --no-cpu-baseline --no-other-callbacks --no-bound-output 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('theta_v=$th 1024x500 hessian', round(d['ms_per_step'],3), d['config']['sweep_terms'], d['config']['outputs_finite'])" >> gpurun_out/r04w/ab_theta.log
DTO_ENGINE_LIB=libdto_engine_t.so DTO_THETA_V=$th timeout -k 10 600 python -m pytest "tests/test_gpu_full_size.py::test_config4_per_rank_share_1024_states_500_knots_of_4000" -x -q 2>&1 | tail -1 >> gpurun_out/r04w/ab_theta.log
done
cat gpurun_out/r04w/ab_theta.log
