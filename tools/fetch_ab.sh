#!/bin/bash
R=/root/repo; OUT=$R/gpurun_out/fetch; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in libavddpg_hip_base libavddpg_hip; do
  AVDDPG_HIP_LIB=$R/avddpg_amd/lib/$lib.so rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$lib -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $OUT/$lib -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_avg.py "$f" FETCH_SIZE | head -2
done
rm -rf $OUT
