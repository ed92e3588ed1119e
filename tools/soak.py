#!/usr/bin/env python3
"""Diagnostic: long runs (device RNG, auto-reset) of the fused nofrl path and the batched interfrl learner at 256x5
platoons; prints weight magnitude / finiteness every 300 steps."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avddpg_amd import config, trainer
for kw, fused, steps in ((dict(), True, 1500), (dict(fed_method="interfrl", weighted_average_enabled=False), False, 600), (dict(fed_method="interfrl"), False, 600)):
    conf = config.Config(num_platoons=256, pl_size=5, buffer_size=2000, **kw)
    eng = "batched" if kw else None
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, fused_update=fused, shared_engine=eng)
    vt.reset_episode()
    t0 = time.time()
    for i in range(steps):
        vt.step()
        if i % 300 == 299:
            torch.cuda.synchronize()
            th = vt.agents.theta
            print(kw, fused, i + 1, "finite", bool(torch.isfinite(th).all()), "max|w|", float(th.abs().max()), "episodes", vt.episode,
                  "mean reward/step", float(vt.env.reward.mean()), flush=True)
    print("time", round(time.time() - t0, 1))
