import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from avddpg_amd import config, trainer
res = {}
for engine in ("per_agent", "fused", "batched"):
    conf = config.Config(num_platoons=8, pl_size=3, buffer_size=4096, fed_method="interfrl", weighted_average_enabled=False,
                         episode_sim_time=20.0)
    np.random.seed(21)
    vt = trainer.VecTrainer(conf, rng="host", shared_sets=True, shared_engine=engine)
    ep, avg = vt.run(number_of_episodes=8)
    r = np.array([[ep[p][m] for m in range(3)] for p in range(8)])  # [P, M, episodes]
    res[engine] = r
    print(engine, "steps/episode", conf.steps_per_episode, "mean episodic reward per episode:", np.round(r.mean(axis=(0, 1)), 3))
a = res["per_agent"]
for e in ("fused", "batched"):
    d = np.abs(res[e] - a)
    print(e, "max |diff| per episode:", np.round(d.max(axis=(0, 1)), 4), " rel to |reward|:", np.round(d.max(axis=(0, 1)) / np.abs(a).mean(axis=(0, 1)), 4))
