#!/usr/bin/env python3
"""Diagnostic: time avd_learn_shared_bf16 (csrc/wide.hip) at bench sizes.
usage: time_wide.py [P] [M] [H1] [H2] [Ha] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from avddpg_amd import config, vec

P, M, H1, H2, Ha, iters = (int(x) for x in (sys.argv[1:] + [4096, 5, 256, 128, 48, 5][len(sys.argv) - 1:]))
conf = config.Config(actor_layer1_size=H1, actor_layer2_size=H2, critic_layer1_size=H1, critic_layer2_size=H2,
                     critic_act_layer_size=Ha)
grp = vec.AgentGroup(M, 4, 1, conf)
rows = P * 64
f = lambda *s: torch.randn(*s, device="cuda")
s, a, r, s2 = f(M, rows, 4), f(M, rows, 1), f(M, rows), f(M, rows, 4)
g = grp.learn_shared(s, a, r, s2, P * M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    grp.learn_shared(s, a, r, s2, P * M, grads=g)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
N = M * rows
KC = H1 + Ha
flops = 2.0 * N * H2 * (2 * H1 + 3 * KC)      # forward GEMMs: actor x2 (target, online), critic x3
flops += 2.0 * N * H2 * (KC + H1)              # weight-gradient GEMMs
flops += 2.0 * N * H2 * (KC + Ha + H1)         # input-gradient GEMMs
print(f"P={P} M={M} widths={H1}/{H2}/{Ha}: {ms:.3f} ms per learn  ({flops / ms * 1e-9:.1f} TFLOP/s of GEMM work, "
      f"workspace {grp._wide_ws.numel() / 2**30:.2f} GiB)")
