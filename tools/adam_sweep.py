#!/usr/bin/env python3
"""Diagnostic: adam_polyak_kernel time vs blocks-per-weight-set (AVD_ADAM_GX) at 20480 sets."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avddpg_amd import config, vec
n = 20480
grp = vec.AgentGroup(n, 4, 1, config.Config())
g = torch.randn(n, grp.lay.theta_size, device="cuda") * 1e-3
for gx in sys.argv[1:]:
    os.environ["AVD_ADAM_GX"] = gx
    for _ in range(2):
        grp.apply(g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        grp.apply(g)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"gx={gx:>3s}: {ms:.2f} ms  actual {9 * grp.lay.theta_size * 4 * n / ms / 1e9:.2f} TB/s  algorithmic {2.47e6 * n / ms / 1e9:.2f} TB/s")
