#!/bin/bash
# SQ counters of the config-5 (hidden 1024) learn kernels: tools/c5_pmc.sh [kernel-name substring, default fwd_gen]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/c5_pmc; rm -rf $OUT; mkdir -p $OUT
K=${1:-fwd_gen}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --mode interfrl --hidden 1024 --steps 2 --warmup 1 --no-cpu-baseline"
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "MfmaUtil"; do
  n=$(echo $C | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$n -o run -- $B > /dev/null 2>&1
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$K" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in row["Kernel_Name"]: continue
    rows[(row["Dispatch_Id"], row["Kernel_Name"].split("(")[0][:40])][row["Counter_Name"]] = float(row["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (d, k), v in rows.items():
    for c, x in v.items(): agg[k][c].append(x)
for k, v in agg.items():
    print(k, " ".join(f"{c.replace('SQ_', '')}={sum(x) / len(x):.3e}" for c, x in v.items()))
PY
done
rm -rf $OUT
