#!/bin/bash
# Regenerates the round-6 artefacts of profiles/ in one gpurun call (outputs under gpurun_out/profiles_r06/; copy what is to be judged
# into profiles/). usage: /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/make_profiles_r06.sh'
# Every profiler pass runs bench.py with --prewarm-seconds 0 (the time-based prewarm made `steps` / `warmup` not describe
# what was profiled).
# ONLY="driver config5" (environment) restricts the run to those groups: driver, config5, intrafrl, weighted, rccl, env, headline
R=${GRAFT_REPO_ROOT:-/root/repo}; T=r06; OUT=$R/gpurun_out/profiles_$T; rm -rf $OUT; mkdir -p $OUT; cd $R
want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }
# the driver's own command (all five lines in one JSON) and the 2000-step default
want driver && python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${T}_bench_driver_command.json 2>$OUT/err.txt
want driver && python bench.py > $OUT/${T}_bench_default.json 2>>$OUT/err.txt
# BASELINE configs[4] on its own, longer; intrafrl; weighted vs unweighted on the same box; the one-rank RCCL communicator
want config5 && python bench.py --mode interfrl --hidden 1024 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${T}_bench_config5_hidden1024.json 2>>$OUT/err.txt
want intrafrl && python bench.py --mode intrafrl --no-cpu-baseline --steps 300 --warmup 30 > $OUT/${T}_bench_intrafrl.json 2>>$OUT/err.txt
want intrafrl && python bench.py --mode intrafrl --intra-chunks 1 --no-cpu-baseline --steps 300 --warmup 30 > $OUT/${T}_bench_intrafrl_unpipelined.json 2>>$OUT/err.txt
want intrafrl && python bench.py --mode intrafrl --directional --no-cpu-baseline --steps 300 --warmup 30 > $OUT/${T}_bench_intrafrl_directional.json 2>>$OUT/err.txt
want weighted && python bench.py --mode interfrl --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/${T}_bench_interfrl_unweighted.json 2>>$OUT/err.txt
want weighted && python bench.py --mode interfrl --weighted --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/${T}_bench_interfrl_weighted.json 2>>$OUT/err.txt
want rccl && python bench.py --one-rank-rccl --mode interfrl --no-cpu-baseline --steps 1000 --warmup 200 > $OUT/${T}_bench_one_rank_rccl.json 2>>$OUT/err.txt
want env && python tools/env_bandwidth_fused.py $OUT/${T}_env_bandwidth_step_fused.json > $OUT/${T}_env_bandwidth_step_fused.txt 2>>$OUT/err.txt
cd /tmp && export TMPDIR=/tmp
pmc_pair() {  # tag, then the arguments of tools/pmc_workload.py
  tag=$1; shift
  for cn in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $OUT/pmc_${tag}_$cn -o run -- python3 $R/tools/pmc_workload.py "$@" > /dev/null 2>&1
  done
  python3 $R/tools/pmc_traffic.py "$(find $OUT/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)" "$(find $OUT/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)" $OUT/${T}_pmc_traffic_$tag.json > $OUT/${T}_pmc_traffic_$tag.txt 2>&1
  rm -rf $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE
}
want config5 && pmc_pair interfrl_h1024 interfrl 2 5 1024   # BASELINE configs[4]: the rank-one chain
want intrafrl && pmc_pair intrafrl intrafrl 3
# kernel stats + MfmaUtil: config 5, intrafrl, and the (unchanged) headline chain for this round's box
stats() {  # tag, bench arguments
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -o run -- python3 $R/bench.py --prewarm-seconds 0 --no-cpu-baseline "$@" > $OUT/${T}_bench_under_rocprof_$tag.json 2>/dev/null
  s=$(find $OUT/stats_$tag -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/${T}_kernel_stats_$tag.csv
  rm -rf $OUT/stats_$tag
}
mfma() {  # tag, bench arguments
  tag=$1; shift
  rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_$tag -o run -- python3 $R/bench.py --prewarm-seconds 0 --no-cpu-baseline "$@" > /dev/null 2>&1
  c=$(find $OUT/mfma_$tag -name "*counter_collection.csv" | head -1); [ -n "$c" ] && python3 $R/tools/pmc_avg.py "$c" MfmaUtil $OUT/${T}_mfma_util_$tag.json > /dev/null 2>&1
  rm -rf $OUT/mfma_$tag
}
want config5 && stats config5 --mode interfrl --hidden 1024 --steps 10 --warmup 4
want config5 && mfma config5 --mode interfrl --hidden 1024 --steps 4 --warmup 2
want intrafrl && stats intrafrl --mode intrafrl --steps 30 --warmup 10
want headline && stats interfrl --mode interfrl --steps 60 --warmup 20
cd $R
ls -la $OUT
