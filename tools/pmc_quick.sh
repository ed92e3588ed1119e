#!/bin/bash
# Diagnostic: HBM-side bytes per agent of the default bench kernels (FETCH_SIZE / WRITE_SIZE in separate passes).
# usage (on the GPU box): tools/pmc_quick.sh <tag> [extra bench args]
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
done
cd $R
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python tools/pmc_summary.py "$(f FETCH_SIZE)" "$(f WRITE_SIZE)" $OUT/pmc_traffic.json 2>&1 | grep -i "learn\|adam\|correction\|rows"
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
