#!/bin/bash
# package power and clock while the config-5 learner runs (run from the repo root on the GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; }
python bench.py --mode interfrl --hidden 1024 --steps 1200 --warmup 5 --no-cpu-baseline > /tmp/p5.log 2>/dev/null &
PID=$!; sleep 14; for i in 1 2 3 4; do smi; sleep 1; done; wait $PID; head -c 220 /tmp/p5.log; echo
