#!/bin/bash
# package power and clock while the gemm-like ping-pong probe runs on constant and on random operands (bf16, and the same bytes as fp16)
R=${GRAFT_REPO_ROOT:-/root/repo}
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; }
for bin in lds_dma_bw lds_dma_bw_f16; do
  [ -x $R/tools/probes/$bin ] || continue
  for mode in const random; do
    echo "== $bin, operands: $mode"
    if [ $mode = random ]; then export PROBE_RANDOM=1; else unset PROBE_RANDOM; fi
    PROBE_LOOP=300 $R/tools/probes/$bin > /tmp/pd.log 2>&1 &
    PID=$!; sleep 2.0; smi; sleep 0.7; smi; wait $PID; tail -1 /tmp/pd.log
  done
done
