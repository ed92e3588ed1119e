#!/bin/bash
# idle time between consecutive kernels of the default interfrl step (kernel trace): tools/gap_trace.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/gap; mkdir -p $R/gpurun_out/gap
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gap -o run -- python3 $R/bench.py --mode interfrl --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$R/gpurun_out/gap/run_kernel_trace.csv'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find the last 'step_fused' kernel and take the kernels after the previous one
idx=[i for i,r in enumerate(rows) if 'step_fused' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
prev_end=None; tot_k=0; tot_gap=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1e3 if prev_end else 0
    print(f"{r['Kernel_Name'].split('(')[0][-60:]:60s} dur {(e-s)/1e3:8.1f} us  gap before {gap:6.1f}")
    tot_k+=(e-s)/1e3; tot_gap+=gap; prev_end=e
print('kernels', round(tot_k,1), 'gaps', round(tot_gap,1), 'step', round((int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3,1))
PY
rm -rf $R/gpurun_out/gap
