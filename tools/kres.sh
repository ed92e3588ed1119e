#!/bin/bash
# Kernel resource usage (VGPRs, scratch, LDS, occupancy) of one csrc file as the Makefile builds it.  usage: tools/kres.sh fsplit.hip [extra flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; F=$1; shift
X=""; case $F in wide.hip) X="-mllvm -amdgpu-mfma-vgpr-form";; fset.hip) X="-fno-honor-nans";; fsplit.hip) X="-fno-honor-nans -fno-slp-vectorize";; esac
cd $R/avddpg_amd/csrc && /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function $X "$@" -x hip -c $F -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c "
import sys,re,subprocess
cur=None
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass',l)
    if not m:
        if 'warning' in l or 'error' in l: print(l.rstrip())
        continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        name=t.split(': ')[1]
        try: name=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',name],capture_output=True,text=True).stdout.strip()
        except Exception: pass
        cur={'name':re.sub(r'\(.*','',name)}
    elif cur is not None:
        k,v=t.split(': ')
        cur[k.strip()]=v
        if k.startswith('LDS Size'):
            print(f\"{cur['name'][:78]:78s} vgpr {cur.get('VGPRs','?'):>4} agpr {cur.get('AGPRs','?'):>3} sgpr {cur.get('TotalSGPRs','?'):>3} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4} occ {cur.get('Occupancy [waves/SIMD]','?')} lds {v}\")
"
