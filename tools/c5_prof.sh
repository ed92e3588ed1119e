#!/bin/bash
# per-kernel table of BASELINE config 5 (hidden 1024): tools/c5_prof.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/c5
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5 -o run -- python3 $R/bench.py --mode interfrl --hidden 1024 --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/c5/bench.json 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/c5/run_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:10.1f} us  per-step {float(r['TotalDurationNs'])/7/1e6:7.2f} ms")
PY
head -c 300 $R/gpurun_out/c5/bench.json
