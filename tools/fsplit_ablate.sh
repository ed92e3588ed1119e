#!/bin/bash
# Diagnostic: where dw_kernel of the split set learner spends its cycles -- the kernel alone (AVD_FSPLIT_ONLY=dw) in the
# -DAVD_STAMP build, which accumulates s_memtime deltas per loop phase and wave (fetch issue / compute / stage / barrier):
#   tools/fsplit_ablate.sh build   (here: compiles avddpg_amd/lib/libavddpg_hip_abl_STAMP.so)
#   tools/fsplit_ablate.sh run dw  (on the GPU box; AVD_FSPLIT_ONLY also takes head and dx for plain timings)
# r03 findings: a one-tile-deep prefetch was NOT the limit (two tiles deep: no change); the two waves of a SIMD ran their split
# VALU and their MFMAs in lockstep -- 5400 cycles per tile for 2750 of MFMA -- until the units were software-pipelined.
R=$(cd $(dirname $0)/.. && pwd); C=$R/avddpg_amd/csrc
VARS="STAMP"
if [ "$1" = build ]; then
  for v in $VARS; do
    /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -fno-honor-nans -fno-slp-vectorize -DAVD_$v -x hip -c $C/fsplit.hip -o $C/build/_abl.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $C/build/capi.cpp.o $C/build/env.hip.o $C/build/replay.hip.o $C/build/mlp.hip.o $C/build/lean.hip.o $C/build/optim.hip.o $C/build/wide.hip.o $C/build/fset.hip.o $C/build/_abl.o $C/build/act.hip.o -o $R/avddpg_amd/lib/libavddpg_hip_abl_$v.so
  done
else
  K=${2:-dw}
  for i in 1 2; do
    AVD_FSPLIT_ONLY=$K python $R/tools/fsplit_time.py 2>/dev/null | tail -1
    for v in $VARS; do AVD_FSPLIT_ONLY=$K AVDDPG_HIP_LIB=$R/avddpg_amd/lib/libavddpg_hip_abl_$v.so python $R/tools/fsplit_time.py 2>/dev/null | tail -1; done
  done
fi
