#!/usr/bin/env python3
"""Diagnostic: avd_learn_set_split_f16x3 (csrc/fsplit.hip) -- per-tensor error against the exact-f32 per-agent kernel +
federated mean (and the bf16 set learner beside it) on perturbed weights, then timing at bench size.
usage: fsplit_check.py [P] [M] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from avddpg_amd import config, vec
from tests.gpu_util import t
from tests.test_gpu_mlp import _perturbed_group

P, M, iters = (int(x) for x in (sys.argv[1:] + [4096, 5, 10][len(sys.argv) - 1:]))
names = ["cWs", "cbs", "cWa", "cba", "cgs", "cbes", "cga", "cbea", "cW2", "cb2", "cg3", "cbe3", "cW3", "cb3",
         "aW1", "ab1", "ag1", "abe1", "aW2", "ab2", "ag2", "abe2", "aW3", "ab3"]
for p_, m_ in ((8, 3), (70, 5)):
    conf, grp = _perturbed_group(m_, S=4, seed=71)
    rs = np.random.RandomState(72)
    n, B = p_ * m_, 64
    s = rs.normal(0, 1.5, size=(n, B, 4)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n, B, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n, B))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n, B, 4)).astype(np.float32)
    avg = vec.fed_mean(grp.learn(t(s), t(a), t(r), t(s2), m_), p_, m_, method=conf.interfrl)
    sp = grp.learn_set_split(t(s), t(a), t(r), t(s2), n)
    bf = grp.learn_set_fused(t(s), t(a), t(r), t(s2), n)
    print(f"P={p_} M={m_}: per-tensor max error / max of the tensor against f32 per-agent kernel + fed_mean (set 0)")
    cs, as_ = grp.grads_as_lists(sp[0])
    cb, ab = grp.grads_as_lists(bf[0])
    ce, ae = grp.grads_as_lists(avg[0])
    for nm, x, y, z in zip(names, cs + as_, cb + ab, ce + ae):
        sc = np.abs(z).max() + 1e-30
        print(f"   {nm:5s} split {np.abs(x - z).max() / sc:.2e}   bf16 {np.abs(y - z).max() / sc:.2e}   max {sc:.3e}")

conf = config.Config()
grp = vec.AgentGroup(M, 4, 1, conf)
n = P * M
f = lambda *sh: torch.randn(*sh, device="cuda")
s, a, r, s2 = 1.5 * f(n, 64, 4), f(n, 64, 1), -f(n, 64).abs() * 0.3, 1.5 * f(n, 64, 4)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = grp.learn_set_split(s, a, r, s2, n)
ms = timed(lambda: grp.learn_set_split(s, a, r, s2, n, grads=g))
flops = 0.751e6 * 64 * n  # 0.751 MFLOP per sample
print(f"split set learner  P={P} M={M}: {ms:.3f} ms per learn ({flops / ms * 1e-9:.1f} TFLOP/s algorithmic, workspace "
      f"{grp._fsplit_ws.numel() / 2**30:.2f} GiB)")
g2 = grp.learn_set_fused(s, a, r, s2, n)
print(f"bf16 set learner   : {timed(lambda: grp.learn_set_fused(s, a, r, s2, n, grads=g2)):.3f} ms")
gp = torch.zeros(n, grp.lay.theta_size, device="cuda")
grp.learn(s, a, r, s2, M, grads=gp)
avg = vec.fed_mean(gp, P, M, method=conf.interfrl)
for name, lo, hi in (("actor", 0, grp.lay.actor_size), ("critic", grp.lay.actor_size, grp.lay.theta_size)):
    sc = avg[:, lo:hi].abs().max().item()
    print(f"  {name}: max |split - f32 mean| / max = {(g[:, lo:hi] - avg[:, lo:hi]).abs().max().item() / sc:.2e}   "
          f"bf16: {(g2[:, lo:hi] - avg[:, lo:hi]).abs().max().item() / sc:.2e}")
