#!/bin/bash
# the config-5 artefacts of profiles/ alone (bench line, per-kernel table, MfmaUtil by kernel): tools/c5_artefacts.sh [tag]
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c5_artefacts
rm -rf $OUT && mkdir -p $OUT
cd $R
python bench.py --mode interfrl --hidden 1024 --no-cpu-baseline --steps 200 --warmup 20 > $OUT/${TAG}_bench_config5_hidden1024.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --mode interfrl --hidden 1024"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma -o run -- $B --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
cp "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_config5.csv
python tools/pmc_avg.py "$(find $OUT/mfma -name '*counter_collection.csv' | head -1)" MfmaUtil $OUT/${TAG}_mfma_util_config5.json | head -8
rm -rf $OUT/stats $OUT/mfma
