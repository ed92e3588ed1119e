R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/regen; rm -rf $OUT; mkdir -p $OUT; cd $R; TAG=r02
python bench.py --mode interfrl --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_per_agent.json 2>/dev/null
python bench.py --mode interfrl --engine batched --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_batched.json 2>/dev/null
python bench.py --mode interfrl --engine fused --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_fused.json 2>/dev/null
python tools/time_fset.py > $OUT/${TAG}_fset_vs_other_learners.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_interfrl_fused -o run -- $B --mode interfrl --engine fused > $OUT/${TAG}_bench_under_rocprof_interfrl_fused.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_interfrl_per_agent -o run -- $B --mode interfrl > $OUT/${TAG}_bench_under_rocprof_interfrl_per_agent.json 2>/dev/null
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_interfrl_fused -o run -- $B --steps 3 --mode interfrl --engine fused > /dev/null 2>&1
cd $R
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python tools/pmc_avg.py "$(f mfma_interfrl_fused)" MfmaUtil $OUT/${TAG}_mfma_util_interfrl_fused.json > /dev/null 2>&1
bash tools/fset_pmc.sh > $OUT/${TAG}_fset_sq_counters.txt 2>/dev/null
for d in stats_interfrl_per_agent stats_interfrl_fused; do
  s=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/${TAG}_kernel_stats_${d#stats_}.csv
done
rm -rf $OUT/stats_* $OUT/mfma_interfrl_fused
ls $OUT
