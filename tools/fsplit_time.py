#!/usr/bin/env python3
"""Diagnostic: time avd_learn_set_split_bf16x3 at bench size (AVDDPG_HIP_LIB selects the library build; tools/ab_fsplit.sh).
usage: fsplit_time.py [P] [M] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from avddpg_amd import config, vec

P, M, iters = (int(x) for x in (sys.argv[1:] + [4096, 5, 20][len(sys.argv) - 1:]))
grp = vec.AgentGroup(M, 4, 1, config.Config())
n = P * M
f = lambda *sh: torch.randn(*sh, device="cuda")
s, a, r, s2 = 1.5 * f(n, 64, 4), f(n, 64, 1), -f(n, 64).abs() * 0.3, 1.5 * f(n, 64, 4)
g = grp.learn_set_split(s, a, r, s2, n)
for _ in range(5):
    grp.learn_set_split(s, a, r, s2, n, grads=g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    grp.learn_set_split(s, a, r, s2, n, grads=g)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('AVDDPG_HIP_LIB', 'default'):50s} split learn {e0.elapsed_time(e1) / iters:.4f} ms  checksum {float(g.double().abs().sum()):.9e}")
