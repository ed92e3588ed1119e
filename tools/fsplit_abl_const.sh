#!/bin/bash
# The timing ablations of the split engine with the ablation COMPILED IN (-DFSPLIT_ABL_CONST=<bits>: no runtime branch in any scheduled
# region), same box, alternating with the product library: time per learn (tools/fsplit_time.py) and per-kernel averages.
# Build the variant libraries first: lib/libavddpg_hip_abl<bits>.so. usage: tools/fsplit_abl_const.sh [bits...]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/fsplit_abl_const.txt
for v in product ${@:-2 4 6}; do
  L=$R/avddpg_amd/lib/libavddpg_hip_abl$v.so; [ $v = product ] && L=$R/avddpg_amd/lib/libavddpg_hip.so
  rm -rf $OUT/ablk
  AVDDPG_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ablk -o run -- python3 $R/tools/fsplit_time.py 200 > $OUT/ablk.log 2>&1
  echo "== $v: $(grep 'us per learn' $OUT/ablk.log | sed 's/.*: //')" >> $OUT/fsplit_abl_const.txt
  s=$(find $OUT/ablk -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && python3 - "$s" >> $OUT/fsplit_abl_const.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if ("dw_kernel" in n or "dx_kernel" in n or "head_kernel" in n) and int(r["Calls"]) >= 100:
        print(f"   {n.split('(')[0].replace('void avd::fsplit::', '')[:52]:52s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
  rm -rf $OUT/ablk
done
cat $OUT/fsplit_abl_const.txt
