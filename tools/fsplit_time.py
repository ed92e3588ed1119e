#!/usr/bin/env python3
"""Time avd_learn_set_split_f16x3 alone at 4096 x 5 (per-launch HIP events over `reps` back-to-back learns after a warm-up) and
print its per-tensor distance from the exact-f32 engine. usage: fsplit_time.py [reps] [L]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from avddpg_amd import config, vec
from tests.test_gpu_fset import NAMES
from tests.test_gpu_mlp import _perturbed_group

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P, B, S = 4096, 64, 4
conf, grp = _perturbed_group(M, S=S, seed=91)
gen = torch.Generator(device="cuda").manual_seed(92)
rn = lambda *sh: torch.randn(*sh, device="cuda", generator=gen)
n = P * M
s, a, r, s2 = 1.5 * rn(n, B, S), 2.5 * (2 * torch.rand(n, B, 1, device="cuda", generator=gen) - 1), -rn(n, B).abs() * 0.3, 1.5 * rn(n, B, S)
g = grp.learn_set_split(s, a, r, s2, n).clone()
exact = vec.fed_mean(grp.learn(s, a, r, s2, M), P, M, method=conf.interfrl)
worst = {}
for k in range(M):
    gs, ge = grp.grads_as_lists(g[k]), grp.grads_as_lists(exact[k])
    for name, x, z in zip(NAMES, gs[0] + gs[1], ge[0] + ge[1]):
        worst[name] = max(worst.get(name, 0.0), float(np.abs(x - z).max() / max(1e-30, np.abs(z).max())))
print("vs exact-f32 engine, worst per tensor:", " ".join(f"{k}:{v:.1e}" for k, v in worst.items()), "finite:", bool(torch.isfinite(g).all()))
for _ in range(50):
    grp.learn_set_split(s, a, r, s2, n)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    grp.learn_set_split(s, a, r, s2, n)
e1.record()
torch.cuda.synchronize()
print(f"learn_set_split 4096 x {M}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per learn over {reps} back-to-back learns")
if os.environ.get("FSPLIT_DETAIL"):
    name = os.environ["FSPLIT_DETAIL"]
    i = NAMES.index(name)
    for k in range(M):
        gs, ge = grp.grads_as_lists(g[k]), grp.grads_as_lists(exact[k])
        x, z = (gs[0] + gs[1])[i], (ge[0] + ge[1])[i]
        d = np.abs(x - z) / np.abs(z).max()
        flat = np.argsort(-d.ravel())[:6]
        print(f"set {k} {name} shape {x.shape}: max rel err {d.max():.1e}; worst elements (index, ref, err):",
              [(tuple(int(v) for v in np.unravel_index(j, x.shape)), f"{z.ravel()[j]:.2e}", f"{(x - z).ravel()[j]:.1e}") for j in flat])
        if x.ndim == 2:
            print("   per input row:", [f"{d[q].max():.1e}" for q in range(x.shape[0])], " features over 1e-5:", int((d.max(axis=0) > 1e-5).sum()), "of", x.shape[1])
