#!/bin/bash
# LDS activity / bank conflicts of the config-5 kernels and of the gemm-like probe (same tile, same reads)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/lds; rm -rf $OUT; mkdir -p $OUT
[ -x $R/tools/probes/lds_dma_bw ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/probes/lds_dma_bw.hip -o $R/tools/probes/lds_dma_bw  # (the binary is not tracked)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/c5 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mode interfrl --hidden 1024 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/probe -o run -- $R/tools/probes/lds_dma_bw > /dev/null 2>&1
cd $R
for d in c5 probe; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:10]:
    a, c, ad = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_ADDR_CONFLICT", 0)
    print(f"{k[:56]:56s} n={cnt[k]:4d} lds_active={a:.3e} bank_conflict={c:.3e} ({100 * c / max(a, 1):.1f}%) addr_conflict={ad:.3e}")
PY
done
rm -rf $OUT/c5 $OUT/probe
