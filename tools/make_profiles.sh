#!/bin/bash
# Regenerates every artefact under profiles/ on the GPU box (one gpurun call): tools/make_profiles.sh <tag>
# Outputs go to gpurun_out/profiles_<tag>/; copy what is to be judged into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
AVD_LEARN_KERNEL=fast python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_default_learn_kernel_t.json 2>/dev/null
python bench.py --no-fused --no-cpu-baseline > $OUT/${TAG}_bench_unfused.json 2>/dev/null
python bench.py --mode interfrl --engine per_agent --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_per_agent.json 2>/dev/null
python bench.py --mode interfrl --engine batched --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_batched.json 2>/dev/null
python bench.py --mode interfrl --engine fused --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_fused.json 2>/dev/null
python tools/time_fset.py > $OUT/${TAG}_fset_vs_other_learners.txt 2>/dev/null
python bench.py --mode interfrl --hidden 1024 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_config5_hidden1024.json 2>/dev/null
python bench.py --pl-size 10 --buffer-size 50000 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_config3_4096x10.json 2>/dev/null
python bench.py --framework centralized --no-cpu-baseline > $OUT/${TAG}_bench_centralized_4096x5.json 2>/dev/null
python tools/phase_profile.py 4096 lean > $OUT/${TAG}_phase_profile_learn_kernel_l.txt 2>/dev/null
python tools/phase_profile.py 4096 lean-fused > $OUT/${TAG}_phase_profile_learn_kernel_l_fused.txt 2>/dev/null
[ -x tools/probes/overlap ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value tools/probes/overlap.hip -o tools/probes/overlap
# model of the fused tile: <lds KB> <0: 8-byte, 1: 16-byte operand accesses, 2: 16-byte + half blocks (3 WG/CU fit)> <tiles>
# <MFMAs per wave in the compute phase> <sleeps per 64 MFMAs; negative: the compute phase only idles> <stream> <rowmap> <in place>
# <start delay of every second workgroup of a CU, x 8128 cycles>
(cd tools/probes && for a in "150 0 20480 4800 2 1 0 0" "150 1 20480 4800 2 1 0 0" "75 1 20480 4800 2 1 0 0" "75 1 20480 4800 2 1 0 1" \
   "75 1 20480 4800 -2 1 0 0" "75 1 20480 4800 2 1 0 0 20" "75 2 20480 4800 2 1 0 0" "50 2 20480 4800 2 1 0 0" "150 1 20480 0 0 1 0 0" "75 1 20480 0 0 1 0 0" "50 2 20480 0 0 1 0 0"; do ./overlap $a; done) > $OUT/${TAG}_overlap_probe.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fused -o run -- $B > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_unfused -o run -- $B --no-fused > $OUT/${TAG}_bench_under_rocprof_unfused.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_interfrl_per_agent -o run -- $B --mode interfrl --engine per_agent > $OUT/${TAG}_bench_under_rocprof_interfrl_per_agent.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_interfrl_fused -o run -- $B --mode interfrl --engine fused > $OUT/${TAG}_bench_under_rocprof_interfrl_fused.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_fused_$c -o run -- $B --steps 3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_unfused_$c -o run -- $B --steps 3 --no-fused > /dev/null 2>&1
done
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_fused -o run -- $B --steps 3 > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_unfused -o run -- $B --steps 3 --no-fused > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_interfrl_fused -o run -- $B --steps 3 --mode interfrl --engine fused > /dev/null 2>&1
cd $R
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python tools/pmc_summary.py "$(f pmc_fused_FETCH_SIZE)" "$(f pmc_fused_WRITE_SIZE)" $OUT/pmc_traffic.json > $OUT/pmc_fused.log 2>&1
python tools/pmc_summary.py "$(f pmc_unfused_FETCH_SIZE)" "$(f pmc_unfused_WRITE_SIZE)" $OUT/pmc_traffic_unfused.json > $OUT/pmc_unfused.log 2>&1
python tools/pmc_avg.py "$(f mfma_fused)" MfmaUtil $OUT/${TAG}_mfma_util_fused.json > /dev/null 2>&1
python tools/pmc_avg.py "$(f mfma_unfused)" MfmaUtil $OUT/${TAG}_mfma_util_unfused.json > /dev/null 2>&1
python tools/pmc_avg.py "$(f mfma_interfrl_fused)" MfmaUtil $OUT/${TAG}_mfma_util_interfrl_fused.json > /dev/null 2>&1
bash tools/fset_pmc.sh > $OUT/${TAG}_fset_sq_counters.txt 2>/dev/null
for d in stats_fused stats_unfused stats_interfrl_per_agent stats_interfrl_fused; do
  s=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/${TAG}_kernel_stats_${d#stats_}.csv
done
# the raw per-dispatch traces are large: keep the summaries only
rm -rf $OUT/stats_* $OUT/pmc_fused_* $OUT/pmc_unfused_* $OUT/mfma_fused $OUT/mfma_unfused $OUT/mfma_interfrl_fused
ls -la $OUT
