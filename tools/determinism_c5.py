#!/usr/bin/env python3
"""Race detector for the wide learner at BASELINE config 5's size: the same learn_shared call N times on the same inputs; beyond
f32 atomics' summation order (~1e-6 of a slab's max) every repeat must agree. Prints the worst deviation per slab.
Usage: tools/determinism_c5.py [repeats] [platoons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_configs_full import _wide_group

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
M, B, S = 5, 64, 4
conf, grp = _wide_group(M, 121)
g = torch.Generator(device="cuda").manual_seed(122)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
s, a = 1.5 * rn(M, P * B, S), 2.5 * (2 * torch.rand(M, P * B, 1, device="cuda", generator=g) - 1)
r, s2 = -rn(M, P * B).abs() * 0.3, 1.5 * rn(M, P * B, S)
lay = grp.lay
print({k: getattr(lay, k) for k in dir(lay) if not k.startswith('_') and isinstance(getattr(lay, k), int)})
ref = grp.learn_shared(s, a, r, s2, P * M).clone()
worst = [0.0, 0.0]
bad = 0
for i in range(N):
    out = grp.learn_shared(s, a, r, s2, P * M)
    for j, (lo, hi) in enumerate(((0, lay.actor_size), (lay.actor_size, lay.theta_size))):
        d = ((out[:, lo:hi] - ref[:, lo:hi]).abs().max() / ref[:, lo:hi].abs().max()).item()
        worst[j] = max(worst[j], d)
        bad += d > 1e-4
        if d > 1e-4:
            dd = (out[:, lo:hi] - ref[:, lo:hi]).abs()
            sets = (dd.max(dim=1).values / ref[:, lo:hi].abs().max()).tolist()
            bad_idx = (dd > 1e-4 * ref[:, lo:hi].abs().max()).nonzero()
            cols = bad_idx[:, 1] + lo
            import collections
            hist = collections.Counter((int(c) // 1024) for c in cols.tolist())
            print(f"repeat {i}: slab {'actor' if j == 0 else 'critic'} deviates {d:.2e}; per set {['%.1e' % x for x in sets]}; "
                  f"{len(cols)} elements in [{int(cols.min())}, {int(cols.max())}]; per 1024-block: {sorted(hist.items())[:12]}")
print(f"worst deviation over {N} repeats: actor {worst[0]:.2e}, critic {worst[1]:.2e}; slabs off: {bad}")
