#!/bin/bash
# Same-box A/B of two builds of the library on the split learner: alternates `tools/fsplit_time.py` between the product library and
# AVDDPG_HIP_LIB=<other> (default lib/libavddpg_hip_base.so), N rounds.  usage: tools/ab_lib.sh [other.so] [rounds] [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-$R/avddpg_amd/lib/libavddpg_hip_base.so}; N=${2:-3}; REPS=${3:-400}
for i in $(seq $N); do
  echo -n "new : "; python3 $R/tools/fsplit_time.py $REPS 2>/dev/null | tail -1
  echo -n "base: "; AVDDPG_HIP_LIB=$B python3 $R/tools/fsplit_time.py $REPS 2>/dev/null | tail -1
done
