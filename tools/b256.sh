#!/bin/bash
# 256 x 2000 (and 128 x 1000) Hessian / Jacobian with the product library: ms per call, dominant kernel, sweep (run through gpurun)
for rep in 1 2; do
for spec in "256 2000 hessian 10" "256 2000 jacobian 10" "128 1000 hessian 20" "128 1000 jacobian 20" "256 250 jacobian 20"; do set -- $spec
timeout -k 10 400 python bench.py --states $1 --knots $2 --callback $3 --steps $4 --warmup 2 --no-cpu-baseline --no-other-callbacks --no-bound-output 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 x $2 $3', round(d['ms_per_step'],3), 'dominant', round(d['roofline']['avg_launch_ms'],3), 'sweep', round(d.get('secondary_kernel',{}).get('ms_per_step',0),3), d['config']['outputs_finite'])"
done; done
