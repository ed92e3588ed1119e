#!/usr/bin/env python3
"""Bandwidth-regime data point of the kernel that actually runs the environment in the throughput mode (SURVEY
section 7, hard part 1): step_fused_kernel (avd_step_fused_f32: OU noise -> policy clip -> leader exog -> platoon step -> replay
add -> reward counters, one launch) at P = 2^12 .. 2^21 platoons x 5 vehicles. At BASELINE's P = 4096 a step moves 2 MB and is
launch-latency bound; the HBM roofline is read at P = 2^20. Algorithmic bytes per vehicle-step of the FUSED step: read x 16 +
prev_a 4 + actor output 4 + OU state 4 + reward counter 4, write x' 16 + prev_a 4 + reward 4 + action 4 + OU state 4 + reward
counter 4 + terminal flag 1 + replay row 40 = 109 B, + 5 B per platoon (leader exog, done) = 550 B per platoon-step at L = 5
(SURVEY 8(d)'s 245 B is the environment step alone); without the replay add (ring = NULL) 69 B per vehicle-step = 350 B.  usage: env_bandwidth_fused.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from avddpg_amd import config, trainer

L, out = 5, []
for logp in (12, 16, 20, 21):
    P = 1 << logp
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=4, fed_method="interfrl", weighted_average_enabled=False)
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, shared_sets=True, shared_engine="fused3")  # (5 weight sets, no per-agent slab)
    vt.reset_episode()
    vt.actor_out.uniform_(-2.5, 2.5)
    for with_replay in (True, False):
        if not with_replay:
            vt.replay.ring = None  # avd_step_fused_f32 with ring = NULL: everything but the replay add (the ring's 40-byte rows lie
                                   # `capacity` rows apart per agent: 5 M scattered partial-line writes at P = 2^20 -- a property of the
                                   # [agent][capacity][row] layout, invisible at P = 4096)
        for _ in range(5):
            vt._step_fused()
        n = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            vt._step_fused()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        per = (109 if with_replay else 69) * L + 5
        out.append({"kernel": "step_fused_kernel" + ("" if with_replay else " (ring = NULL: no replay add)"), "P": P, "L": L, "us_per_step": us,
                    "platoon_steps_per_s": P / us * 1e6, "algorithmic_bytes_per_platoon_step": per, "algorithmic_GBps": per * P / us / 1e3,
                    "frac_of_8TBps": per * P / us / 1e3 / 8000})
        print(out[-1], flush=True)
    del vt
    torch.cuda.empty_cache()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "env_bandwidth_fused.json")
json.dump(out, open(path, "w"), indent=1)
