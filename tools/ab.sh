#!/bin/bash
# A/B of two library builds in ONE gpurun call (boxes differ by several percent): tools/ab.sh libA.so libB.so
A=${1:-avddpg_amd/lib/libavddpg_hip_base.so}; B=${2:-avddpg_amd/lib/libavddpg_hip.so}
for i in 1 2; do python tools/time_learn.py $A 2>/dev/null | tail -1; python tools/time_learn.py $B 2>/dev/null | tail -1; done
