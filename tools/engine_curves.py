#!/usr/bin/env python3
"""Diagnostic: episodic-reward curves of the shared-set engines against the exact-f32 engine (per_agent: f32 per-agent learn kernel
+ federated sum) on the SAME host RNG stream (parity mode): 8 platoons x 3 vehicles, interfrl, 200-step episodes, ~1500 updates per
weight set. What an engine's gradient error does to the quantity the reference plots (workers/trainer.py:510-517).
usage: engine_curves.py [episodes]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from avddpg_amd import config, trainer

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res, th = {}, {}
for engine in ("per_agent", "fused3", "fused", "batched"):
    conf = config.Config(num_platoons=8, pl_size=3, buffer_size=4096, fed_method="interfrl", weighted_average_enabled=False,
                         episode_sim_time=20.0)
    np.random.seed(21)
    vt = trainer.VecTrainer(conf, rng="host", shared_sets=True, shared_engine=engine)
    ep, avg = vt.run(number_of_episodes=episodes)
    r = np.array([[ep[p][m] for m in range(3)] for p in range(8)])  # [P, M, episodes]
    res[engine], th[engine] = r, vt.agents.theta.cpu().numpy()
    print(f"{engine:9s} steps/episode {conf.steps_per_episode}, updates per set {int(vt.agents.step[0])}; mean episodic reward per episode:",
          np.round(r.mean(axis=(0, 1)), 3))
a = res["per_agent"]
print("max |episodic reward - exact engine's| over the 24 agents, per episode (and relative to the mean |reward|):")
for e in ("fused3", "fused", "batched"):
    d = np.abs(res[e] - a)
    print(f"  {e:8s}", np.round(d.max(axis=(0, 1)), 4), " rel", np.round(d.max(axis=(0, 1)) / np.abs(a).mean(axis=(0, 1)), 5))
    dt = np.abs(th[e] - th["per_agent"])
    print(f"           weights after the run: mean |theta - exact| {dt.mean():.3e}, max {dt.max():.3e}")
