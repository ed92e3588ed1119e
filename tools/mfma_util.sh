#!/bin/bash
# MFMA utilisation (rocprofv3 derived counter MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM)) per kernel,
# own PMC pass as MI355X_MICROARCH.md prescribes: tools/mfma_util.sh <tag>
TAG=${1:-r01}; R=/root/repo; OUT=$R/gpurun_out/mfma_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/fused -o run -- $B > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/unfused -o run -- $B --no-fused > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/config5 -o run -- $B --mode interfrl --hidden 1024 > /dev/null 2>&1
cd $R
for d in fused unfused config5; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  echo "== $d"; python tools/pmc_avg.py "$f" MfmaUtil $OUT/${TAG}_mfma_util_$d.json | head -8
done
rm -rf $OUT/fused $OUT/unfused $OUT/config5
