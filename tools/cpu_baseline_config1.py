#!/usr/bin/env python3
"""SURVEY 8(d)'s CPU line: the reference-shaped loop (oracle/trainer.py: per-object Python loops of workers/trainer.py:251-356, NumPy
float64 environment, float32 networks) on BASELINE configs[0] -- 1 platoon x 3 vehicles -- for >= 5000 training steps on ONE thread
(the reference's own setting, src/rand.py:14-15), plus the all-cores figure: N independent single-thread copies (platoons are
independent), N = the cores this process may run on. TensorFlow is not installable here, so this is a port ("kind": "port"), not
the reference itself. usage: cpu_baseline_config1.py [steps] [seconds for the all-cores figure]"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(steps=None, seconds=None, q=None):
    from threadpoolctl import threadpool_limits

    from oracle import platoon, trainer

    with threadpool_limits(limits=1):
        tr = trainer.RefTrainer(platoon.EnvParams(), 1, 3, seed=1, buffer_size=100000, fed_method="normal")
        tr.reset_episode()
        for _ in range(65):
            if tr.step():
                tr.reset_episode()
        n, t0, u0 = 0, time.perf_counter(), tr.updates
        while (steps is not None and n < steps) or (seconds is not None and time.perf_counter() - t0 < seconds):
            if tr.step():
                tr.reset_episode()
            n += 1
        dt = time.perf_counter() - t0
    out = (n / dt, (tr.updates - u0) / dt, n, dt)
    if q is not None:
        q.put(out)
    return out


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    v, u, n, dt = run(steps=steps)
    res = {"config": "BASELINE configs[0]: 1 platoon x 3 vehicles, DDPG nofrl, 1 update per env step after the 65-step replay warm-up",
           "one_thread": {"env_steps_per_s": v, "agent_updates_per_s": u, "steps": n, "seconds": dt, "cores": 1, "kind": "port"}}
    ncpu = len(os.sched_getaffinity(0))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=run, kwargs=dict(seconds=secs, q=q)) for _ in range(ncpu)]
    for p in procs:
        p.start()
    got = []
    deadline = time.monotonic() + 3 * secs + 120
    while len(got) < ncpu and time.monotonic() < deadline:
        try:
            got.append(q.get(timeout=1.0))
        except Exception:
            if not any(p.is_alive() for p in procs):
                break
    for p in procs:
        p.join(timeout=2)
        if p.is_alive():
            p.terminate()
    res["all_cores"] = {"env_steps_per_s": sum(g[0] for g in got), "agent_updates_per_s": sum(g[1] for g in got), "processes": len(got),
                        "host_cores": ncpu, "seconds_each": secs,
                        "note": "independent single-thread copies of the same loop, throughputs summed (platoons are independent)"}
    print(json.dumps(res, indent=1))
