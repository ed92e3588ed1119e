#!/bin/bash
# rocprofv3 kernel stats of the shared-set learner: tools/prof_wide.sh P M H1 H2 Ha iters tag
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_$7
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$7 -o wide -- python3 /root/repo/tools/time_wide.py $1 $2 $3 $4 $5 $6 > /dev/null 2>&1
python3 /root/repo/tools/prof_top.py /root/repo/gpurun_out/prof_$7 14
