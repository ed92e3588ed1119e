#!/bin/bash
# the pieces tools/make_profiles.sh does not cover: learn_kernel_l phase stamps (diagnostic build) and config 5's MfmaUtil by kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/extras_r03
rm -rf $OUT && mkdir -p $OUT
cd $R
python tools/phase_profile.py 4096 lean > $OUT/r03_phase_profile_learn_kernel_l.txt 2>/dev/null
python tools/phase_profile.py 4096 lean-fused > $OUT/r03_phase_profile_learn_kernel_l_fused.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_config5 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mode interfrl --hidden 1024 > /dev/null 2>&1
cd $R
python tools/pmc_avg.py "$(find $OUT/mfma_config5 -name '*counter_collection.csv' | head -1)" MfmaUtil $OUT/r03_mfma_util_config5.json
rm -rf $OUT/mfma_config5
