#!/usr/bin/env python3
"""Diagnostic: how far the shared-set weights of the bf16 engines drift from the exact-f32 engine's over the first updates
(same host RNG stream), per tensor, in units of lr * updates (Adam moves every weight by <= lr per step)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from avddpg_amd import config, trainer

P, L, steps = 16, 3, 64 + int(sys.argv[1]) if len(sys.argv) > 1 else 64 + 20
runs = {}
for engine in ("per_agent", "fused", "batched"):
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=4096, fed_method="interfrl", weighted_average_enabled=False)
    np.random.seed(11)
    vt = trainer.VecTrainer(conf, rng="host", shared_sets=True, shared_engine=engine)
    vt.reset_episode()
    for i in range(steps):
        vt.step(0, i)
    runs[engine] = vt
a = runs["per_agent"]
lay = a.agents.lay
n_upd = steps - 64
names = ["aW1", "ab1", "ag1", "abe1", "aW2", "ab2", "ag2", "abe2", "aW3", "ab3"]
cn = ["cWs", "cbs", "cgs", "cbes", "cWa", "cba", "cga", "cbea", "cW2", "cb2", "cg3", "cbe3", "cW3", "cb3"]
offs = [(n, getattr(lay, n), conf.actor_lr) for n in names] + [(n, lay.actor_size + getattr(lay, n), conf.critic_lr) for n in cn]
offs.sort(key=lambda x: x[1])
ends = [o[1] for o in offs[1:]] + [lay.theta_size]
print(f"updates = {n_upd}; drift = mean |theta - theta_f32| / (lr * updates)   [max]")
for (n, lo, lr), hi in zip(offs, ends):
    row = []
    for e in ("fused", "batched"):
        d = (runs[e].agents.theta[:, lo:hi] - a.agents.theta[:, lo:hi]).abs() / (lr * n_upd)
        row.append(f"{e} {d.mean().item():.4f} [{d.max().item():.3f}]")
    print(f"{n:5s} {'  '.join(row)}")
