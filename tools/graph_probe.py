#!/usr/bin/env python3
"""Diagnostic: what a captured HIP graph of one training step would buy (timing only -- the Philox / replay counters are
host-side scalars baked into the captured launches, so a replayed graph repeats the same random numbers).
usage: graph_probe.py [platoons] [mode: nofrl|fused]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from avddpg_amd import config, trainer

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
kw = dict(fed_method="interfrl", weighted_average_enabled=False) if mode == "fused" else {}
conf = config.Config(num_platoons=P, pl_size=5, buffer_size=2000, episode_sim_time=1e7, **kw)
vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, fused_update=(mode != "fused"),
                        shared_engine="fused" if mode == "fused" else None)
vt.replay.buffer_counter = 2000
vt.reset_episode()
for _ in range(5):
    vt.step()
torch.cuda.synchronize()


def timed(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timed(vt.step)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        vt.step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    vt.step()
graph = timed(g.replay)
print(f"P={P} mode={mode}: eager {eager:.3f} ms/step, captured graph replay {graph:.3f} ms/step")
