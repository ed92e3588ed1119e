#!/usr/bin/env python3
"""Diagnostic: long runs (device RNG, auto-reset) of the fused nofrl path and the batched interfrl learner at 256x5
platoons (layer-wise, fused and per-agent engines); prints weight magnitude / finiteness every 300 steps."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avddpg_amd import config, trainer
for kw, fused, eng, steps in ((dict(), True, None, 1500),
                             (dict(fed_method="interfrl", weighted_average_enabled=False), False, "batched", 600),
                             (dict(fed_method="interfrl", weighted_average_enabled=False), False, "fused", 3000),
                             (dict(fed_method="interfrl", weighted_average_enabled=False), False, "per_agent", 600)):
    conf = config.Config(num_platoons=256, pl_size=5, buffer_size=2000, **kw)
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, fused_update=fused, shared_engine=eng)
    vt.reset_episode()
    t0 = time.time()
    for i in range(steps):
        vt.step()
        if i % 300 == 299:
            torch.cuda.synchronize()
            th = vt.agents.theta
            print(kw, fused, eng, i + 1, "finite", bool(torch.isfinite(th).all()), "max|w|", float(th.abs().max()), "episodes", vt.episode,
                  "mean reward/step", float(vt.env.reward.mean()), flush=True)
    print("time", round(time.time() - t0, 1))
# weighted federated averaging needs the per-episode reward history: the episode loop (run), past the weighting window
conf = config.Config(num_platoons=16, pl_size=3, buffer_size=500, fed_method="interfrl", weighted_average_enabled=True,
                     weighted_window=3, total_time_steps=8 * 600)
vt = trainer.VecTrainer(conf, rng="device", auto_reset=False)
ep, avg = vt.run()
th = vt.agents.theta
print("weighted interfrl via run():", len(ep[0][0]), "episodes, finite", bool(torch.isfinite(th).all()), "max|w|", float(th.abs().max()),
      "weights used from episode", conf.weighted_window, "last avg reward", float(avg[0][0][-1]))
