#!/bin/bash
# package power of instruction mixes around v_mfma_f32_32x32x16_f16 on toggling operands (tools/probes/mfma_mix), 2 waves per SIMD:
# MFMA alone, + 6 VALU per MFMA, + 0.75 ds_read_b128 per MFMA, + both
R=${GRAFT_REPO_ROOT:-/root/repo}
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; }
for cfg in "0 0" "3 0" "0 3" "3 3"; do
  MFMA_ITERS=20000000 $R/tools/probes/mfma_mix $cfg 512 > /tmp/pm.log 2>&1 &
  PID=$!; sleep 1.6; smi; sleep 0.8; smi; wait $PID; cat /tmp/pm.log
done
