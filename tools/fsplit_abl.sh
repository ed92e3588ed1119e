#!/bin/bash
# Timing ablations of the split engine's dw / dx kernels (DIAGNOSTIC library, results wrong by design): AVD_FSPLIT_ABL bit 1 = every
# tile fetch reads the workgroup's first tile (L2 hits, no HBM traffic), bit 2 = no workgroup barrier in the tile loop, bit 4 = the relu / split (heads, dw) and BN-ReLU-backward (dx) VALU skipped.
# Per variant: per-kernel averages under rocprofv3 --kernel-trace --stats of tools/fsplit_time.py (4096 x 5).
# usage: tools/fsplit_abl.sh [variants...]   (default: 0 1 2 3)   -> gpurun_out/fsplit_abl.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AVDDPG_HIP_LIB=$R/avddpg_amd/lib/libavddpg_hip_diag.so
V=${@:-0 1 2 3 4}
: > $OUT/fsplit_abl.txt
for v in $V; do
  export AVD_FSPLIT_ABL=$v
  rm -rf $OUT/abl_raw
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abl_raw -o run -- python3 $R/tools/fsplit_time.py 120 > $OUT/abl_$v.log 2>&1
  s=$(find $OUT/abl_raw -name "*kernel_stats.csv" | head -1)
  echo "== AVD_FSPLIT_ABL=$v: $(grep 'us per learn' $OUT/abl_$v.log)" >> $OUT/fsplit_abl.txt
  [ -n "$s" ] && python3 - "$s" >> $OUT/fsplit_abl.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if ("dw_kernel" in n or "dx_kernel" in n or "dxa" in n or "head_kernel" in n) and int(r["Calls"]) >= 100:
        print(f"   {n[:70]:70s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
  rm -rf $OUT/abl_raw
done
cat $OUT/fsplit_abl.txt
