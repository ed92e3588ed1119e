cd /tmp && export TMPDIR=/tmp
for e in base exp1 exp2 exp4 exp8 exp15; do
  if [ $e = base ]; then unset AVDDPG_HIP_LIB; else export AVDDPG_HIP_LIB=$GRAFT_REPO_ROOT/avddpg_amd/lib/$e.so; fi
  rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats -d /tmp/pf -o fs -- python3 $GRAFT_REPO_ROOT/tools/time_fset.py 4096 5 10 > /dev/null 2>&1
  echo "== $e"; python3 $GRAFT_REPO_ROOT/tools/prof_top.py /tmp/pf 2>&1 | grep head_kernel | sed 's/void avd::fset:://' | cut -c1-40,70-120
done
