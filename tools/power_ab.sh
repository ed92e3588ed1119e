#!/bin/bash
# Package power and shader clock (rocm-smi) while the split learner runs back to back, for two builds of the library alternately:
# is the chain at the power cap, and does a build whose kernels idle less run at a lower clock? usage: tools/power_ab.sh libA.so libB.so
R=${GRAFT_REPO_ROOT:-/root/repo}
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; }
for round in 1 2; do for L in $1 $2; do
  echo "== $(basename $L)"
  AVDDPG_HIP_LIB=$R/$L python3 $R/tools/fsplit_time.py 2500 > /tmp/pab.log 2>&1 &
  PID=$!; sleep 5; for i in 1 2 3; do smi; sleep 0.7; done; wait $PID; grep "us per learn" /tmp/pab.log | sed 's/.*: //'
done; done
