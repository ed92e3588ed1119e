#!/usr/bin/env python3
"""Diagnostic: time avd_learn_set_fused_bf16 (csrc/fset.hip) at bench sizes, beside the layer-wise learner (wide.hip) and the
per-agent f32 kernel with shared sets; prints each launch's share when run under rocprofv3.
usage: time_fset.py [P] [M] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from avddpg_amd import config, vec

P, M, iters = (int(x) for x in (sys.argv[1:] + [4096, 5, 10][len(sys.argv) - 1:]))
conf = config.Config()
grp = vec.AgentGroup(M, 4, 1, conf)
n = P * M
f = lambda *s: torch.randn(*s, device="cuda")
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


g = grp.learn_set_fused(s, a, r, s2, n)
ms = timed(lambda: grp.learn_set_fused(s, a, r, s2, n, grads=g))
flops = 0.751e6 * 64 * n  # 0.751 MFLOP per sample
print(f"fused set learner   P={P} M={M}: {ms:.3f} ms per learn  ({flops / ms * 1e-9:.1f} TFLOP/s algorithmic, "
      f"workspace {grp._fset_ws.numel() / 2**30:.2f} GiB)")
if "--all" in sys.argv or len(sys.argv) < 2:
    sm = lambda x: x.view(P, M, *x.shape[1:]).transpose(0, 1).reshape(M, P * 64, *x.shape[2:]).contiguous()
    ws = sm(s), sm(a), sm(r), sm(s2)
    gw = grp.learn_shared(*ws, n)
    print(f"layer-wise learner  : {timed(lambda: grp.learn_shared(*ws, n, grads=gw)):.3f} ms")
    d = (g - gw).abs().max().item() / gw.abs().max().item()
    print(f"max |fused - layerwise| / max = {d:.2e}")
    gp = torch.zeros(n, grp.lay.theta_size, device="cuda")
    print(f"per-agent f32 kernel: {timed(lambda: grp.learn(s, a, r, s2, M, grads=gp)):.3f} ms (+ fed_sum)")
    avg = vec.fed_mean(gp, P, M, method=conf.interfrl)
    for name, lo, hi in (("actor", 0, grp.lay.actor_size), ("critic", grp.lay.actor_size, grp.lay.theta_size)):
        sc = avg[:, lo:hi].abs().max().item()
        print(f"  {name}: max |fused - f32 mean| / max = {(g[:, lo:hi] - avg[:, lo:hi]).abs().max().item() / sc:.2e}   "
              f"layer-wise: {(gw[:, lo:hi] - avg[:, lo:hi]).abs().max().item() / sc:.2e}")
