#!/bin/bash
# Builds avddpg_amd/lib/libavddpg_hip_phase.so: the diagnostic library with per-phase cycle stamps in the per-agent learn kernels
# (-DAVD_DIAG -DAVD_PHASE_TIMING; tools/phase_profile.py loads it). Own object directory, per-file flags as in the Makefile.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/avddpg_amd/csrc; mkdir -p build_phase
F="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -DAVD_DIAG -DAVD_PHASE_TIMING"
for f in capi.cpp env.hip replay.hip mlp.hip lean.hip optim.hip wide.hip fset.hip fsplit.hip act.hip cen.hip; do
  X=""; case $f in wide.hip) X="-mllvm -amdgpu-mfma-vgpr-form";; fset.hip) X="-fno-honor-nans";; fsplit.hip) X="-fno-honor-nans -fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc $F $X -x hip -c $f -o build_phase/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build_phase/*.o -o ../lib/libavddpg_hip_phase.so && echo built
