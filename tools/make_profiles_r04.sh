#!/bin/bash
# Regenerates the round-4 artefacts of profiles/ in one gpurun call (outputs under gpurun_out/profiles_r04/; copy what is to be judged
# into profiles/). usage: /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/make_profiles_r04.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; T=r04; OUT=$R/gpurun_out/profiles_$T; rm -rf $OUT; mkdir -p $OUT; cd $R
python bench.py > $OUT/${T}_bench_default.json 2>$OUT/err.txt
python bench.py --pl-size 10 --buffer-size 50000 --no-cpu-baseline --steps 400 --warmup 50 > $OUT/${T}_bench_config3_4096x10.json 2>>$OUT/err.txt
python bench.py --framework centralized --no-cpu-baseline --steps 300 --warmup 30 > $OUT/${T}_bench_centralized_4096x5.json 2>>$OUT/err.txt
python bench.py --mode interfrl --hidden 1024 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${T}_bench_config5_hidden1024.json 2>>$OUT/err.txt
python bench.py --mode interfrl --engine per_agent --no-cpu-baseline --steps 300 --warmup 30 > $OUT/${T}_bench_interfrl_per_agent.json 2>>$OUT/err.txt
python bench.py --gpus 2 --backend gloo --single-device --platoons 256 --buffer-size 4096 --steps 50 --warmup 10 --no-cpu-baseline --mode interfrl > $OUT/${T}_bench_2ranks_one_gpu_gloo.json 2>>$OUT/err.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline"
for m in interfrl nofrl; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o run -- $B --mode $m > $OUT/${T}_bench_under_rocprof_$m.json 2>/dev/null
  s=$(find $OUT/stats_$m -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/${T}_kernel_stats_$m.csv
  rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_$m -o run -- $B --steps 6 --warmup 4 --mode $m > /dev/null 2>&1
  c=$(find $OUT/mfma_$m -name "*counter_collection.csv" | head -1); [ -n "$c" ] && python3 $R/tools/pmc_avg.py "$c" MfmaUtil $OUT/${T}_mfma_util_$m.json > /dev/null 2>&1
  for cn in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $OUT/pmc_${m}_$cn -o run -- python3 $R/tools/pmc_workload.py $m 3 > /dev/null 2>&1
  done
  f() { find $OUT/pmc_${m}_$1 -name "*counter_collection.csv" | head -1; }
  python3 $R/tools/pmc_traffic.py "$(f FETCH_SIZE)" "$(f WRITE_SIZE)" $OUT/${T}_pmc_traffic_$m.json > $OUT/${T}_pmc_traffic_$m.txt 2>&1
  rm -rf $OUT/stats_$m $OUT/mfma_$m $OUT/pmc_${m}_FETCH_SIZE $OUT/pmc_${m}_WRITE_SIZE
done
rocprofv3 --kernel-trace --output-format csv -d $OUT/cen_trace -o run -- python3 $R/bench.py --framework centralized --no-cpu-baseline --steps 10 --warmup 5 > /dev/null 2>&1
c=$(find $OUT/cen_trace -name "*kernel_trace.csv" | head -1); [ -n "$c" ] && python3 $R/tools/cen_trace.py "$c" 33 > $OUT/${T}_centralized_update_timeline.txt 2>&1
rm -rf $OUT/cen_trace
for cn in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $OUT/pmc_centralized_$cn -o run -- python3 $R/tools/pmc_workload.py centralized 3 > /dev/null 2>&1
done
python3 $R/tools/pmc_traffic.py "$(find $OUT/pmc_centralized_FETCH_SIZE -name "*counter_collection.csv" | head -1)" "$(find $OUT/pmc_centralized_WRITE_SIZE -name "*counter_collection.csv" | head -1)" $OUT/${T}_pmc_traffic_centralized.json > $OUT/${T}_pmc_traffic_centralized.txt 2>&1
rm -rf $OUT/pmc_centralized_FETCH_SIZE $OUT/pmc_centralized_WRITE_SIZE
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/mfma_cen -o run -- python3 $R/bench.py --framework centralized --no-cpu-baseline --steps 6 --warmup 4 > /dev/null 2>&1
c=$(find $OUT/mfma_cen -name "*counter_collection.csv" | head -1); [ -n "$c" ] && python3 $R/tools/pmc_avg.py "$c" MfmaUtil $OUT/${T}_mfma_util_centralized.json > /dev/null 2>&1
rm -rf $OUT/mfma_cen
cd $R
[ -f avddpg_amd/lib/libavddpg_hip_phase.so ] || bash tools/build_phase_lib.sh > /dev/null 2>&1
(echo "# python tools/phase_profile.py 4096 centralized  (tools/build_phase_lib.sh library: cen::learn_kernel_c, gradients out, 4096 models S = 20, A = 5; shader cycles of wave 0 per model)"; python tools/phase_profile.py 4096 centralized 2>&1 | grep -v "amdgpu.ids\| 0 cyc/tile") > $OUT/${T}_phase_profile_centralized.txt
bash tools/fsplit_pmc.sh > $OUT/${T}_fsplit_kernels_and_sq_counters.txt 2>/dev/null
python tools/cpu_baseline_config1.py > $OUT/${T}_cpu_baseline_config1.txt 2>&1
(python tools/determinism_engines.py 3000; python tools/determinism_engines.py 1500 10) > $OUT/${T}_determinism_of_repeated_learns.txt 2>&1
python tools/fsplit_time.py 1000 > $OUT/${T}_fsplit_time_1000_learns.txt 2>&1
ls -la $OUT
