#!/bin/bash
# Regenerates the round-5 artefacts of profiles/ in one gpurun call (outputs under gpurun_out/profiles_r05/; copy what is to be judged
# into profiles/). usage: /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/make_profiles_r05.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; T=r05; OUT=$R/gpurun_out/profiles_$T; rm -rf $OUT; mkdir -p $OUT; cd $R
python bench.py > $OUT/${T}_bench_default.json 2>$OUT/err.txt
python bench.py --pl-size 10 --buffer-size 100000 --no-cpu-baseline --steps 400 --warmup 50 > $OUT/${T}_bench_config3_4096x10.json 2>>$OUT/err.txt
python bench.py --framework centralized --no-cpu-baseline --steps 300 --warmup 30 > $OUT/${T}_bench_centralized_4096x5.json 2>>$OUT/err.txt
python bench.py --mode interfrl --hidden 1024 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${T}_bench_config5_hidden1024.json 2>>$OUT/err.txt
python bench.py --mode interfrl --engine per_agent --no-cpu-baseline --steps 300 --warmup 30 > $OUT/${T}_bench_interfrl_per_agent.json 2>>$OUT/err.txt
python bench.py --gpus 2 --backend gloo --single-device --platoons 256 --buffer-size 4096 --steps 50 --warmup 10 --no-cpu-baseline --mode interfrl > $OUT/${T}_bench_2ranks_one_gpu_gloo.json 2>>$OUT/err.txt
python bench.py --one-rank-rccl --mode interfrl --no-cpu-baseline --steps 1000 --warmup 200 > $OUT/${T}_bench_one_rank_rccl.json 2>>$OUT/err.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline"
pmc_pair() {  # tag, then the arguments of tools/pmc_workload.py
  tag=$1; shift
  for cn in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $OUT/pmc_${tag}_$cn -o run -- python3 $R/tools/pmc_workload.py "$@" > /dev/null 2>&1
  done
  python3 $R/tools/pmc_traffic.py "$(find $OUT/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)" "$(find $OUT/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)" $OUT/${T}_pmc_traffic_$tag.json > $OUT/${T}_pmc_traffic_$tag.txt 2>&1
  rm -rf $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE
}
for m in interfrl nofrl; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o run -- $B --mode $m > $OUT/${T}_bench_under_rocprof_$m.json 2>/dev/null
  s=$(find $OUT/stats_$m -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/${T}_kernel_stats_$m.csv
  rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_$m -o run -- $B --steps 6 --warmup 4 --mode $m > /dev/null 2>&1
  c=$(find $OUT/mfma_$m -name "*counter_collection.csv" | head -1); [ -n "$c" ] && python3 $R/tools/pmc_avg.py "$c" MfmaUtil $OUT/${T}_mfma_util_$m.json > /dev/null 2>&1
  rm -rf $OUT/stats_$m $OUT/mfma_$m
  pmc_pair $m $m 3
  pmc_pair ${m}_L10 $m 3 10        # BASELINE configs[2]
done
pmc_pair interfrl_h1024 interfrl 2 5 1024   # BASELINE configs[4]
pmc_pair centralized centralized 3
# config 5 under rocprof: kernel stats + MfmaUtil
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o run -- python3 $R/bench.py --mode interfrl --hidden 1024 --steps 10 --warmup 4 --no-cpu-baseline > $OUT/${T}_bench_under_rocprof_config5.json 2>/dev/null
s=$(find $OUT/stats_c5 -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/${T}_kernel_stats_config5.csv
rm -rf $OUT/stats_c5
rocprofv3 --kernel-trace --output-format csv -d $OUT/cen_trace -o run -- python3 $R/bench.py --framework centralized --no-cpu-baseline --steps 10 --warmup 5 > /dev/null 2>&1
c=$(find $OUT/cen_trace -name "*kernel_trace.csv" | head -1); [ -n "$c" ] && python3 $R/tools/cen_trace.py "$c" 33 > $OUT/${T}_centralized_update_timeline.txt 2>&1
rm -rf $OUT/cen_trace
cd $R
bash tools/fsplit_pmc.sh > $OUT/${T}_fsplit_kernels_and_sq_counters.txt 2>/dev/null
bash tools/fsplit_abl.sh > /dev/null 2>&1; cp $R/gpurun_out/fsplit_abl.txt $OUT/${T}_fsplit_dw_dx_ablations.txt
python tools/cpu_baseline_config1.py > $OUT/${T}_cpu_baseline_config1.txt 2>&1
(python tools/determinism_engines.py 2000; python tools/determinism_engines.py 1000 10) > $OUT/${T}_determinism_of_repeated_learns.txt 2>&1
python tools/fsplit_time.py 1000 > $OUT/${T}_fsplit_time_1000_learns.txt 2>&1
ls -la $OUT
