#!/bin/bash
# Regenerates every artefact under profiles/ on the GPU box (one gpurun call): tools/make_profiles.sh <tag>
# Outputs go to gpurun_out/profiles_<tag>/; copy what is to be judged into profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
N="--no-cpu-baseline --steps 200 --warmup 20"
# the driver's line: interfrl (f32-class set learner) as `value`, nofrl under `also_measured`, CPU baseline
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
# sustained figures (>= 2000 timed steps after 200 warm-up)
python bench.py --mode interfrl --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/${TAG}_bench_interfrl_split_sustained_2000.json 2>/dev/null
python bench.py --mode nofrl --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/${TAG}_bench_nofrl_sustained_2000.json 2>/dev/null
# nofrl variants
python bench.py --mode nofrl $N > $OUT/${TAG}_bench_nofrl.json 2>/dev/null
AVDDPG_HIP_LIB=$R/avddpg_amd/lib/libavddpg_hip_diag.so AVD_LEARN_KERNEL=fast python bench.py --allow-diagnostics --mode nofrl $N > $OUT/${TAG}_bench_nofrl_learn_kernel_t.json 2>/dev/null
python bench.py --mode nofrl --no-fused $N > $OUT/${TAG}_bench_nofrl_unfused.json 2>/dev/null
# interfrl engines
python bench.py --mode interfrl --engine per_agent $N > $OUT/${TAG}_bench_interfrl_per_agent.json 2>/dev/null
python bench.py --mode interfrl --engine batched $N > $OUT/${TAG}_bench_interfrl_batched.json 2>/dev/null
python bench.py --mode interfrl --engine fused $N > $OUT/${TAG}_bench_interfrl_fused_bf16.json 2>/dev/null
python bench.py --mode interfrl --engine fused3 $N > $OUT/${TAG}_bench_interfrl_split.json 2>/dev/null
python tools/fsplit_check.py > $OUT/${TAG}_fsplit_accuracy_and_time.txt 2>/dev/null
python tools/time_fset.py > $OUT/${TAG}_fset_vs_other_learners.txt 2>/dev/null
# other BASELINE configs
python bench.py --mode interfrl --hidden 1024 $N > $OUT/${TAG}_bench_config5_hidden1024.json 2>/dev/null
python bench.py --mode nofrl --pl-size 10 --buffer-size 50000 $N --steps 50 --warmup 5 > $OUT/${TAG}_bench_config3_4096x10_nofrl.json 2>/dev/null
python bench.py --mode interfrl --pl-size 10 --buffer-size 50000 $N > $OUT/${TAG}_bench_config3_4096x10_interfrl_split.json 2>/dev/null
python bench.py --mode nofrl --framework centralized $N > $OUT/${TAG}_bench_centralized_4096x5.json 2>/dev/null
# two self-spawned ranks on this one GPU (gloo: RCCL refuses two ranks per device); the same command without the last two flags is
# what runs over RCCL on an N-GPU node
python bench.py --gpus 2 --backend gloo --single-device --platoons 2048 --buffer-size 20000 $N > $OUT/${TAG}_bench_2ranks_one_gpu_gloo.json 2>/dev/null
bash tools/fsplit_pmc.sh > $OUT/${TAG}_fsplit_kernels_and_sq_counters.txt 2>/dev/null
# race detectors: the same learn repeated on the same inputs (every engine at the reference widths; the wide learner at config 5)
(echo "# tools/determinism_engines.py 1000 and tools/determinism_c5.py 1000: the same learn call repeated on the same inputs"; python tools/determinism_engines.py 1000 2>/dev/null | tail -4; python tools/determinism_c5.py 1000 2>/dev/null | grep worst) > $OUT/${TAG}_determinism_of_repeated_learns.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o run -- $B > $OUT/${TAG}_bench_under_rocprof_default.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_interfrl_split -o run -- $B --mode interfrl > $OUT/${TAG}_bench_under_rocprof_interfrl_split.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_nofrl -o run -- $B --mode nofrl > $OUT/${TAG}_bench_under_rocprof_nofrl.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_nofrl_unfused -o run -- $B --mode nofrl --no-fused > $OUT/${TAG}_bench_under_rocprof_nofrl_unfused.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_interfrl_per_agent -o run -- $B --mode interfrl --engine per_agent > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_interfrl_fused_bf16 -o run -- $B --mode interfrl --engine fused > /dev/null 2>&1
# config 5 (hidden 1024): per-kernel table and MfmaUtil
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config5 -o run -- $B --steps 5 --mode interfrl --hidden 1024 > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_config5 -o run -- $B --steps 2 --warmup 1 --mode interfrl --hidden 1024 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_fused_$c -o run -- $B --steps 3 --mode nofrl > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_unfused_$c -o run -- $B --steps 3 --mode nofrl --no-fused > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_split_$c -o run -- $B --steps 3 --mode interfrl > /dev/null 2>&1
done
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_nofrl -o run -- $B --steps 3 --mode nofrl > /dev/null 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_split -o run -- $B --steps 3 --mode interfrl > /dev/null 2>&1
cd $R
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python tools/pmc_summary.py "$(f pmc_fused_FETCH_SIZE)" "$(f pmc_fused_WRITE_SIZE)" $OUT/pmc_traffic.json > $OUT/pmc_fused.log 2>&1
python tools/pmc_summary.py "$(f pmc_unfused_FETCH_SIZE)" "$(f pmc_unfused_WRITE_SIZE)" $OUT/pmc_traffic_unfused.json > $OUT/pmc_unfused.log 2>&1
python tools/pmc_summary.py "$(f pmc_split_FETCH_SIZE)" "$(f pmc_split_WRITE_SIZE)" $OUT/pmc_traffic_interfrl_split.json > $OUT/pmc_split.log 2>&1
python tools/pmc_avg.py "$(f mfma_nofrl)" MfmaUtil $OUT/${TAG}_mfma_util_nofrl.json > /dev/null 2>&1
python tools/pmc_avg.py "$(f mfma_split)" MfmaUtil $OUT/${TAG}_mfma_util_interfrl_split.json > /dev/null 2>&1
python tools/pmc_avg.py "$(f mfma_config5)" MfmaUtil $OUT/${TAG}_mfma_util_config5.json > /dev/null 2>&1
for d in stats_default stats_interfrl_split stats_nofrl stats_nofrl_unfused stats_interfrl_per_agent stats_interfrl_fused_bf16 stats_config5; do
  s=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/${TAG}_kernel_stats_${d#stats_}.csv
done
# the raw per-dispatch traces are large: keep the summaries only
rm -rf $OUT/stats_* $OUT/pmc_fused_* $OUT/pmc_unfused_* $OUT/pmc_split_* $OUT/mfma_nofrl $OUT/mfma_split $OUT/mfma_config5
ls -la $OUT
