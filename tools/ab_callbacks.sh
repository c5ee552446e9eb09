#!/bin/bash
# A/B of engine BUILDS on the other callbacks inside one gpurun call: tools/ab_callbacks.sh libdto_engine_x.so [...]
# (the named builds live next to libdto_engine.so; the default build is interleaved, 3 repetitions)
run() { DTO_ENGINE_LIB=$1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-callbacks --callback $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', round(d['ms_per_step'],3))"; }
for rep in 1 2 3; do
  for cb in constraint hessian; do
    run libdto_engine.so $cb
    for lib in "$@"; do run $lib $cb; done
  done
done
