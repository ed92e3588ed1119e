#!/bin/bash
# Is the chain power-limited? rocm-smi samples of clock and package power (a) while the split set learner runs back to back,
# (b) while nofrl's learn kernel runs, (c) while a register-resident MFMA loop runs (tools/probes/mfma_dep), (d) idle.
R=${GRAFT_REPO_ROOT:-/root/repo}
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; }
echo "== (a) split set learner (avd_learn_set_split_f16x3), 4096 x 5, 3000 learns back to back"
python $R/tools/fsplit_time.py 4096 5 3000 > /tmp/pp.log 2>&1 &
PID=$!; sleep 4; for i in 1 2 3; do smi; sleep 1; done; wait $PID; tail -1 /tmp/pp.log
echo "== (b) nofrl step (learn_kernel_l), 4096 x 5"
python $R/bench.py --mode nofrl --steps 2500 --warmup 20 --no-cpu-baseline > /tmp/pp2.log 2>/dev/null &
PID=$!; sleep 28; for i in 1 2 3; do smi; sleep 1; done; wait $PID; head -c 200 /tmp/pp2.log; echo
echo "== (c) v_mfma_f32_32x32x16_f16 on register-resident operands, ~1 s per line"
MFMA_ITERS=4000000 $R/tools/probes/mfma_dep > /tmp/pp3.log 2>&1 &
PID=$!; sleep 3; for i in 1 2 3; do smi; sleep 1; done; wait $PID; cat /tmp/pp3.log
echo "== (d) idle"; sleep 2; smi
