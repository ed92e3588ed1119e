#!/bin/bash
# A/B of two library builds on the split set learner in ONE gpurun call (boxes differ by several percent):
# tools/ab_fsplit.sh libA.so libB.so
A=${1:-avddpg_amd/lib/libavddpg_hip_base.so}; B=${2:-avddpg_amd/lib/libavddpg_hip.so}
for i in 1 2 3; do AVDDPG_HIP_LIB=$A python tools/fsplit_time.py 2>/dev/null | tail -1; AVDDPG_HIP_LIB=$B python tools/fsplit_time.py 2>/dev/null | tail -1; done
