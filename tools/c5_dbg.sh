#!/bin/bash
# fwd_gen_kernel with pieces switched off (AVD_FW_DBG bits; results wrong): average kernel time per variant.
# Needs the diagnostics library: hipcc ... -DAVD_FW_DBG -c wide.hip, linked as avddpg_amd/lib/libavddpg_hip_fwdbg.so
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AVDDPG_HIP_LIB=$R/avddpg_amd/lib/libavddpg_hip_fwdbg.so
for d in ${@:-0 1 2 4 8 16 32 63}; do
  rm -rf $R/gpurun_out/c5d; mkdir -p $R/gpurun_out/c5d
  AVD_FW_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5d -o run -- python3 $R/bench.py --mode interfrl --hidden 1024 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "dbg=$d $(grep fwd_gen $R/gpurun_out/c5d/run_kernel_stats.csv | awk -F, '{printf "%s %.0f us  ", substr($1,1,40), $4/1000}')"
done
rm -rf $R/gpurun_out/c5d
