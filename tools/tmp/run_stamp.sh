cd $GRAFT_REPO_ROOT
AVDDPG_HIP_LIB=$GRAFT_REPO_ROOT/avddpg_amd/lib/libavddpg_hip_stamp.so python bench.py --mode interfrl --hidden 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --prewarm-seconds 0 --allow-diagnostics > gpurun_out/stamp.log 2>&1
grep "fwd_gen" gpurun_out/stamp.log | grep -v "wave [1-35-7]" | cut -c1-1500
tail -5 gpurun_out/stamp.log | cut -c1-300
