#!/bin/bash
# config-5 kernel table with an alternative library: tools/c5_probe.sh <lib.so>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AVDDPG_HIP_LIB=$R/$1
mkdir -p $R/gpurun_out/c5p
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5p -o run -- python3 $R/bench.py --mode interfrl --hidden 1024 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/c5p/bench.json 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/c5p/run_kernel_stats.csv')))
for r in rows[:12]:
    if True:
        print(f"{r['Name'][:60]:60s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:10.1f} us")
PY
