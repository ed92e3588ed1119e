#!/bin/bash
# The PMC traffic passes of tools/make_profiles_r04.sh alone (FETCH_SIZE and WRITE_SIZE in separate runs, calibrated on the 1 GiB
# elementwise kernel of the same pass). usage: pmc_only.sh [modes: interfrl nofrl centralized]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=r04; OUT=$R/gpurun_out/profiles_$T; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in ${@:-interfrl nofrl centralized}; do
  for cn in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $OUT/pmc_${m}_$cn -o run -- python3 $R/tools/pmc_workload.py $m 3 > /dev/null 2>&1
  done
  f() { find $OUT/pmc_${m}_$1 -name "*counter_collection.csv" | head -1; }
  python3 $R/tools/pmc_traffic.py "$(f FETCH_SIZE)" "$(f WRITE_SIZE)" $OUT/${T}_pmc_traffic_$m.json > $OUT/${T}_pmc_traffic_$m.txt 2>&1
  rm -rf $OUT/pmc_${m}_FETCH_SIZE $OUT/pmc_${m}_WRITE_SIZE
done
cat $OUT/${T}_pmc_traffic_*.txt
