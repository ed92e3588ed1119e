#!/bin/bash
# Same-box A/B of two builds of the library on the split learner alone: alternately, N rounds, tools/fsplit_time.py (400 learns each).
# usage: tools/ab_two_libs.sh <libA.so> <libB.so> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    echo -n "$(basename $L): "; AVDDPG_HIP_LIB=$R/$L python3 $R/tools/fsplit_time.py 400 2>&1 | grep "us per learn" | sed 's/.*: //'
  done
done
