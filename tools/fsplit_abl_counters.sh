#!/bin/bash
# SQ wave-cycle counters of dx / dw under the timing ablations of tools/fsplit_abl.sh (diagnostic library, results wrong by design):
# does WAIT_ANY go away when the memory (AVD_FSPLIT_ABL=1), the barrier (=2) or both (=3) are taken out? -> gpurun_out/fsplit_abl_counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AVDDPG_HIP_LIB=$R/avddpg_amd/lib/libavddpg_hip_diag.so
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
: > $OUT/fsplit_abl_counters.txt
for v in 0 1 2 3; do
  export AVD_FSPLIT_ABL=$v
  rm -rf $OUT/ablc
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/ablc -o run -- python3 $R/tools/fsplit_check.py 4096 5 4 > /dev/null 2>&1
  f=$(find $OUT/ablc -name "*counter_collection.csv" | head -1)
  echo "== AVD_FSPLIT_ABL=$v (1: every tile fetch from L2, 2: no workgroup barrier in the tile loop, 3: both)" >> $OUT/fsplit_abl_counters.txt
  python3 - "$f" >> $OUT/fsplit_abl_counters.txt <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
for row in csv.DictReader(open(sys.argv[1])):
    if "fsplit" not in row["Kernel_Name"]: continue
    rows[(row["Dispatch_Id"], row["Kernel_Name"].split("(")[0].replace("void avd::fsplit::", ""))][row["Counter_Name"]] = float(row["Counter_Value"])
best = {}
for (d, k), v in rows.items():
    if k not in best or v.get("SQ_WAVE_CYCLES", 0) > best[k].get("SQ_WAVE_CYCLES", 0): best[k] = v
for k, v in sorted(best.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "dx_kernel" not in k and "dw_kernel" not in k: continue
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"   {k[:40]:40s} WAVE_CYCLES {wc:10.3e}  WAIT_ANY {100 * v.get('SQ_WAIT_ANY', 0) / wc:5.1f} %  WAIT_INST_ANY {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f} %  "
          f"ACTIVE_INST_ANY {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f} %  mfma_busy/busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(v.get('SQ_BUSY_CYCLES', 1), 1):6.2f}")
PY
  rm -rf $OUT/ablc
done
cat $OUT/fsplit_abl_counters.txt
