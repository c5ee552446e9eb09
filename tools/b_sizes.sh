#!/bin/bash
# A few shapes with the product library: ms per call and the per-launch times of the chain's kernel families (run through gpurun)
run() { timeout -k 10 400 python bench.py --states $1 --knots $2 --callback $3 --steps ${4:-5} --warmup 2 --no-cpu-baseline --no-other-callbacks --no-bound-output 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'].get('template_instances',{}); print('$1 x $2 $3', round(d['ms_per_step'],3), {k:round(v['avg_launch_ms'],4) for k,v in r.items()}, d['config']['outputs_finite'])"; }
run 1024 500 jacobian 3; run 1024 500 hessian 3; run 512 500 jacobian; run 512 500 hessian; run 256 2000 jacobian 10; run 256 2000 hessian 10; run 128 1000 jacobian 10
