#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -- python3 $R/bench.py --no-cpu-baseline --no-other-callbacks --steps 5 --warmup 2 --callback hessian > /tmp/ph.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/ph/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
tail -1 /tmp/ph.log | cut -c1-200
