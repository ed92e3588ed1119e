#!/bin/bash
# Per-kernel durations + SQ counter pass over the split-operand set learner (tools/fsplit_check.py): tools/fsplit_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/fsplit_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o run -- python3 $R/tools/fsplit_check.py 4096 5 10 > /dev/null 2>&1
f=$(find $OUT/s -name "*kernel_stats.csv" | head -1)
echo "== kernel durations (us, average over launches; 12 launches incl. 1 small-P warm-up pair) =="
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "fsplit" in row["Name"] or "finalize" in row["Name"]:
        print(f'{row["Name"].split("(")[0].replace("void avd::", "")[:60]:60s} calls {row["Calls"]:>4s}  avg {float(row["AverageNs"]) / 1e3:8.1f}  max {float(row["MaxNs"]) / 1e3:8.1f}')
PY
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p -o run -- python3 $R/tools/fsplit_check.py 4096 5 4 > /dev/null 2>&1
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
echo "== SQ counters per launch (largest launches only) =="
python3 - "$f" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
for row in csv.DictReader(open(sys.argv[1])):
    if "fsplit" not in row["Kernel_Name"]: continue
    rows[(row["Dispatch_Id"], row["Kernel_Name"].split("(")[0].replace("void avd::fsplit::", ""))][row["Counter_Name"]] = float(row["Counter_Value"])
best = {}
for (d, k), v in rows.items():
    if k not in best or v.get("SQ_WAVE_CYCLES", 0) > best[k].get("SQ_WAVE_CYCLES", 0): best[k] = v
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]
print("kernel".ljust(40), *[n.replace("SQ_", "")[:13].rjust(14) for n in names], "  mfma_busy/busy")
for k, v in sorted(best.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "kernel" not in k: continue
    print(k[:40].ljust(40), *[f"{v.get(m, 0):14.3e}" for m in names], f"  {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(v.get('SQ_BUSY_CYCLES', 1), 1):.3f}")
PY
rm -rf $OUT/p $OUT/s
