#!/bin/bash
# Every bench line of the round in one gpurun call: tools/bench_all.sh <tag>  -> gpurun_out/bench_<tag>/*.json
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/bench_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/err_default.txt
AVDDPG_HIP_LIB=$R/avddpg_amd/lib/libavddpg_hip_diag.so AVD_LEARN_KERNEL=fast python bench.py --allow-diagnostics --no-cpu-baseline > $OUT/${TAG}_bench_default_learn_kernel_t.json 2>/dev/null
python bench.py --no-fused --no-cpu-baseline > $OUT/${TAG}_bench_unfused.json 2>/dev/null
python bench.py --mode interfrl --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_per_agent.json 2>/dev/null
python bench.py --mode interfrl --engine batched --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_batched.json 2>/dev/null
python bench.py --mode interfrl --engine fused --no-cpu-baseline > $OUT/${TAG}_bench_interfrl_fused.json 2>/dev/null
python bench.py --mode interfrl --hidden 1024 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_config5_hidden1024.json 2>/dev/null
python bench.py --pl-size 10 --buffer-size 50000 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_config3_4096x10.json 2>/dev/null
python bench.py --framework centralized --no-cpu-baseline > $OUT/${TAG}_bench_centralized_4096x5.json 2>/dev/null
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "%.0f env-steps/s  %.2f ms/step  stages=%s  roof=%s %.3f"%(d["value"], d["ms_per_step"], {k:round(v,2) for k,v in d["stages_ms"].items() if v}, d["roofline"]["bound"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
