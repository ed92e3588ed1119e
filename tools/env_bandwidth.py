#!/usr/bin/env python3
"""Bandwidth-regime data point for the environment kernels (hard part 1): at BASELINE's
P = 4096 one env step moves 1 MB and is launch-latency bound, so the HBM roofline of env_step_kernel is measured
at P = 2^20..2^22 platoons x 5 vehicles. Algorithmic bytes: 48 B/vehicle-step + 5 B/platoon."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avddpg_amd import config, vec
L = 5
out = []
for logp in (12, 16, 20, 22):
    P = 1 << logp
    conf = config.Config(pl_size=L)
    env = vec.VecPlatoon(P, L, conf, rng="device")
    env.reset()
    u = torch.rand(P, L, device="cuda") * 5 - 2.5
    ex = torch.randn(P, device="cuda") * 0.1
    for _ in range(5):
        env.step(u, ex)
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        env.step(u, ex)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    alg = (48 * L + 5) * P            # x, prev_a, u read; x', prev_a', reward written; exog, done
    act = alg + L * P                 # + term bytes written by this build
    out.append({"P": P, "L": L, "us_per_step": us, "platoon_steps_per_s": P / us * 1e6,
                "algorithmic_GBps": alg / us / 1e3, "actual_GBps": act / us / 1e3, "frac_of_8TBps": alg / us / 1e3 / 8000})
    print(out[-1])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "r01_env_bandwidth.json"), "w"), indent=1)
