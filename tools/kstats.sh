#!/bin/bash
# Per-kernel durations of the interfrl (fused3) bench step under rocprofv3 --kernel-trace --stats: prints the chain's kernels (avg us).
# usage: tools/kstats.sh [out-tag]   -> gpurun_out/kstats/<tag>_kernel_stats.csv + a printed table
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-run}; OUT=$R/gpurun_out/kstats; mkdir -p $OUT; rm -rf $OUT/raw_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$TAG -o run -- python3 $R/bench.py --mode interfrl --steps 60 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
s=$(find $OUT/raw_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp "$s" $OUT/${TAG}_kernel_stats.csv && python3 - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n, avg, calls = r["Name"], float(r["AverageNs"]) / 1e3, int(r["Calls"])
    if calls >= 60 and ("fsplit" in n or "finalize" in n or "replay" in n or "actor_set" in n or "step_fused" in n or "adam" in n or "polyak" in n or "reset" in n):
        per_step = avg * calls / 80.0
        tot += per_step
        print(f"{n[:86]:86s} calls/step {calls / 80.0:4.1f} avg {avg:8.1f} us  per step {per_step:8.1f}")
print(f"sum per step {tot:.1f} us")
PY
rm -rf $OUT/raw_$TAG
