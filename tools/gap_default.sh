#!/bin/bash
# idle time between consecutive kernels of the default (interfrl split) step: tools/gap_default.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/gapd
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gapd -o run -- python3 $R/bench.py --mode interfrl --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/gapd/bench.json 2>/dev/null
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open('$R/gpurun_out/gapd/run_kernel_trace.csv'))]
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-48:]) for r in rows)
# the timed region: last 30 steps: find step_fused_kernel occurrences
idx=[i for i,e in enumerate(ev) if 'step_fused_kernel' in e[2]]
i0,i1=idx[-21],idx[-1]
seg=ev[i0:i1]
busy=sum(e[1]-e[0] for e in seg); wall=seg[-1][1]-seg[0][0]
print(f"20 steps: wall {wall/20e3:.1f} us/step, kernels busy {busy/20e3:.1f} us/step, idle {(wall-busy)/20e3:.1f} us/step, {len(seg)/20:.1f} launches/step")
gaps=collections.defaultdict(list)
for a,b in zip(seg,seg[1:]): gaps[(a[2][-28:],b[2][-28:])].append(b[0]-a[1])
for k,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1]))[:14]: print(f"{k[0]:>28s} -> {k[1]:<28s} n={len(v)/20:4.1f}/step  mean gap {sum(v)/len(v)/1e3:6.2f} us")
PY
