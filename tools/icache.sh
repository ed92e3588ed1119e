#!/bin/bash
R=/root/repo; OUT=$R/gpurun_out/ic; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in "" "--no-fused"; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS --kernel-trace --output-format csv -d $OUT/r -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $mode > /dev/null 2>&1
f=$(find $OUT/r -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    if "learn_kernel" in k:
        print(k[:60], {c: f"{x:.3e}" for c, x in v.items()}, "miss rate %.2f%%" % (100 * v.get("SQC_ICACHE_MISSES", 0) / max(1, v.get("SQC_ICACHE_REQ", 1))))
PY
rm -rf $OUT/r
done
