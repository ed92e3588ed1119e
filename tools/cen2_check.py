#!/usr/bin/env python3
"""cen2::learn_kernel_c2 (diagnostic switch AVD_CEN2=1 of the diagnostic build) against cen::learn_kernel_c on the same inputs:
per-tensor gradient differences, losses; optional timing. usage: cen2_check.py [n_models] [time]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from avddpg_amd import _hip

_hip.LIB_PATH = os.path.join(ROOT, "avddpg_amd", "lib", "libavddpg_hip_diag.so")
from avddpg_amd import config, vec

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
for L in (5, 3):
    S, A = 4 * L, L
    g = torch.Generator(device="cuda").manual_seed(3)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    grp = vec.AgentGroup(n, S, A, config.Config(), seed=5, hidd_mult=1.2)
    grp.theta.add_(rn(*grp.theta.shape) * 0.02 * (grp.theta != 0))
    grp.theta_t.copy_(grp.theta + rn(*grp.theta.shape) * 0.01 * (grp.theta != 0))
    grp.stats.add_(rn(*grp.stats.shape).abs() * 0.1)
    grp.stats_t.copy_(grp.stats)
    s, a = 1.5 * rn(n, 64, S), 2.5 * (2 * torch.rand(n, 64, A, device="cuda", generator=g) - 1)
    r, s2 = -rn(n, 64).abs() * 0.3, 1.5 * rn(n, 64, S)
    out = {}
    for which in ("cen", "cen2"):
        if which == "cen2":
            os.environ["AVD_CEN2"] = "1"
        else:
            os.environ.pop("AVD_CEN2", None)
        losses = torch.zeros(n, 2, device="cuda")
        gr = grp.learn(s, a, r, s2, 0, losses=losses)
        torch.cuda.synchronize()
        out[which] = (gr.clone(), losses.clone())
    ga, gb = out["cen"][0], out["cen2"][0]
    print(f"L={L}: losses max diff {float((out['cen'][1] - out['cen2'][1]).abs().max()):.2e}; finite {bool(torch.isfinite(gb).all())}")
    worst = 0.0
    for v in range(min(n, 3)):
        la, lb = grp.grads_as_lists(ga[v]), grp.grads_as_lists(gb[v])
        for x, y in zip(la[0] + la[1], lb[0] + lb[1]):
            d = float(np.max(np.abs(np.asarray(x) - np.asarray(y))) / max(1e-12, np.max(np.abs(np.asarray(x)))))
            worst = max(worst, d)
            if d > 3e-6:
                print("   tensor shape", np.asarray(x).shape, "rel diff", f"{d:.2e}")
    print(f"   worst per-tensor relative difference over {min(n, 3)} models: {worst:.2e}; padding equal: {bool(torch.equal(ga == 0, gb == 0))}")
# fused update under cen2 == learn + apply, bit for bit (three updates, 300 models)
os.environ["AVD_CEN2"] = "1"
runs = []
for mode in ("update", "separate"):
    torch.manual_seed(0)
    grp = vec.AgentGroup(300, 20, 5, config.Config(), seed=5, hidd_mult=1.2)
    gg = torch.Generator(device="cuda").manual_seed(9)
    grp.theta.add_(torch.randn(*grp.theta.shape, device="cuda", generator=gg) * 0.02 * (grp.theta != 0))
    grp.theta_t.copy_(grp.theta)
    scratch = torch.empty(300, grp.lay.theta_size, device="cuda")
    losses = torch.zeros(300, 2, device="cuda")
    for k in range(3):
        b = [torch.randn(300, 64, 20, device="cuda", generator=gg), torch.randn(300, 64, 5, device="cuda", generator=gg),
             -torch.rand(300, 64, device="cuda", generator=gg), torch.randn(300, 64, 20, device="cuda", generator=gg)]
        if mode == "update":
            grp.learn_update(*b, scratch, losses=losses)
        else:
            grp.apply(grp.learn(*b, 0, losses=losses))
    torch.cuda.synchronize()
    runs.append([x.clone() for x in (grp.theta, grp.theta_t, grp.stats_t, grp.m, grp.v, losses)])
print("cen2 fused update == learn + apply, bitwise:", [bool(torch.equal(x, y)) for x, y in zip(*runs)])
if len(sys.argv) > 2:
    n = 4096
    grp = vec.AgentGroup(n, 20, 5, config.Config(), seed=5, hidd_mult=1.2)
    f = lambda *s: torch.randn(*s, device="cuda")
    s, a, r, s2 = f(n, 64, 20), f(n, 64, 5), f(n, 64), f(n, 64, 20)
    for which in ("cen", "cen2"):
        if which == "cen2":
            os.environ["AVD_CEN2"] = "1"
        else:
            os.environ.pop("AVD_CEN2", None)
        for _ in range(20):
            grp.learn(s, a, r, s2, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            grp.learn(s, a, r, s2, 0)
        torch.cuda.synchronize()
        print(f"{which}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per learn of 4096 models")
        scratch = torch.empty(n, grp.lay.theta_size, device="cuda")
        for _ in range(10):
            grp.learn_update(s, a, r, s2, scratch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            grp.learn_update(s, a, r, s2, scratch)
        torch.cuda.synchronize()
        print(f"{which}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per learn + update of 4096 models")
