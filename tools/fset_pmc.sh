#!/bin/bash
# SQ counter pass over the fused set learner (tools/time_fset.py): wave-cycle breakdown per kernel. tools/fset_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/fset_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p -o run -- python3 $R/tools/time_fset.py 4096 5 4 > /dev/null 2>&1
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("avd::fset::", "")
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]
print("kernel".ljust(44), *[n.replace("SQ_", "")[:14].rjust(15) for n in names])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if "kernel" not in k: continue
    n = max(cnt[k], 1)
    print(k[:44].ljust(44), *[f"{v[m] / n:15.3e}" for m in names])
PY
rm -rf $OUT/p
