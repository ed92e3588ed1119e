#!/usr/bin/env python3
"""Diagnostic (r04): per-tensor error of the split learner and of the exact-f32 engine against the float64 oracle on the batch and
weights a 4096 x L VecTrainer(fused3) holds at its 70th step (nearly untrained networks: actor gradients ~1e-8).
usage: r04_trainer_state_errors.py [L] [per-feature tensor name]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from avddpg_amd import config, trainer, vec
from oracle import mlp as omlp
from tests.test_gpu_fset import NAMES
from tests.test_gpu_mlp import _nets, _relerr

L = int(sys.argv[1]) if len(sys.argv) > 1 else 5
focus = sys.argv[2] if len(sys.argv) > 2 else "ag1"
P, B = 4096, 64
b = trainer.VecTrainer(config.Config(num_platoons=P, pl_size=L, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False),
                       rng="device", auto_reset=True, seed=7, shared_engine="fused3")
b.reset_episode()
for i in range(69):
    b.step()
spots = (0, L - 1)
pre = {k: _nets(b.agents, k, np.float64) for k in spots}
held = [x.clone() for x in (b.agents.theta, b.agents.stats, b.agents.theta_t, b.agents.stats_t)]
b.step()
torch.cuda.synchronize()
rp = b.replay
ex = vec.AgentGroup(L, b.S, b.A, b.conf, seed=1)
for dst, src in zip((ex.theta, ex.stats, ex.theta_t, ex.stats_t), held):
    dst.copy_(src)
exact = vec.fed_mean(ex.learn(rp.s, rp.a, rp.r, rp.s2, L), P, L, method=b.conf.interfrl)
split = ex.learn_set_split(rp.s, rp.a, rp.r, rp.s2, P * L).clone()
assert torch.equal(split, b.set_grads), "the trainer's gradient is not the learner's on the same inputs"
sn, an, rn, s2n = (x.cpu().numpy() for x in (rp.s, rp.a, rp.r, rp.s2))
for k in spots:
    sel = np.arange(P) * L + k
    cat = lambda x: x[sel].reshape(P * B, *x.shape[2:])
    cg, ag, _ = omlp.learn((cat(sn), cat(an), cat(rn)[:, None], cat(s2n)), *pre[k])
    scg, sag = ex.grads_as_lists(split[k])
    ecg, eag = ex.grads_as_lists(exact[k])
    print(f"set {k}: tensor  split / exact-f32 engine error (of the tensor's max), max |ref|")
    for name, got, eng, ref in zip(NAMES, scg + sag, ecg + eag, cg + ag):
        print(f"  {name:5s} {_relerr(got, ref):.2e} / {_relerr(eng, ref):.2e}   {np.abs(ref).max():.2e}")
        if name == focus:
            d = (got - ref).ravel()
            o = np.argsort(-np.abs(d))[:8]
            print("    worst elements:", [(int(i), f"{ref.ravel()[i]:.3e}", f"{d[i]:.1e}", f"{(eng.ravel()[i] - ref.ravel()[i]):.1e}") for i in o])
            print(f"    error rms {np.sqrt((d ** 2).mean()):.2e}, mean {d.mean():.2e}; ref rms {np.sqrt((ref ** 2).mean()):.2e}")
