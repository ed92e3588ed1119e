#!/bin/bash
# LDS bank-conflict share per kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, own PMC pass): tools/lds_conflicts.sh
R=/root/repo; OUT=$R/gpurun_out/lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/bench -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/c5 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mode interfrl --hidden 1024 > /dev/null 2>&1
cd $R
for d in bench c5; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:8]:
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
    print(f"{k[:64]:64s} lds_active={a:.3e} conflict_cycles={c:.3e} share={100 * c / max(a, 1):.1f}%")
PY
done
rm -rf $OUT/bench $OUT/c5
