cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -4
python bench.py --mode interfrl --hidden 1024 --steps 40 --warmup 5 --no-cpu-baseline --no-extra-configs 2>&1 | tail -1 | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/c5k; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/c5k
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5k -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode interfrl --hidden 1024 --steps 10 --warmup 4 --no-cpu-baseline --no-extra-configs --prewarm-seconds 0 > /dev/null 2>&1
head -12 $GRAFT_REPO_ROOT/gpurun_out/c5k/run_kernel_stats.csv | cut -c1-140
