#!/bin/bash
# the three callbacks at 64 x 1000 with the product library (run through gpurun from the repo root)
for cb in jacobian hessian constraint; do
timeout -k 10 200 python bench.py --states 64 --knots 1000 --callback $cb --no-cpu-baseline --no-other-callbacks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'][20:45], round(d['ms_per_step'],3), d['config']['outputs_finite'], d['config']['sweep_terms'], 'sweep', round(d.get('secondary_kernel',{}).get('ms_per_step',0),3), 'dominant', round(d['roofline']['avg_launch_ms'],3), 'serial', round((d['roofline'].get('timed_region') or {}).get('ms_per_step_serial_pass',0),3))"
done
