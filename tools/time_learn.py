#!/usr/bin/env python3
"""Diagnostic A/B timer: average duration of avd_learn_f32 and avd_learn_update_f32 at bench size for a given build
of the library (`python tools/time_learn.py [path/to/lib.so] [n_agents] [iters]`). Run two builds back to back in the
same gpurun call: boxes differ by several percent. Not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from avddpg_amd import _hip

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
from avddpg_amd import config, vec

n = int(sys.argv[2]) if len(sys.argv) > 2 else 20480
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
conf = config.Config()
grp = vec.AgentGroup(n, 4, 1, conf)
f = lambda *s: torch.randn(*s, device="cuda")
s, a, r, s2 = f(n, 64, 4), f(n, 64, 1), f(n, 64), f(n, 64, 4)
grads = torch.zeros(n, grp.lay.theta_size, device="cuda")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t_learn = timed(lambda: grp.learn(s, a, r, s2, 0, grads=grads))
t_fused = timed(lambda: grp.learn_update(s, a, r, s2, grads))
print(f"{os.path.basename(_hip.LIB_PATH)}: learn {t_learn:.3f} ms  learn+update(fused) {t_fused:.3f} ms  (n={n})")
