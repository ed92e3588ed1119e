#!/usr/bin/env python3
"""Race detector for the learners at the reference widths (4096 platoons x 5 sets): each engine's learn N times on the same
inputs (optionally [sets per platoon], 10 = BASELINE config 3). The set learners (fset.hip / fsplit.hip) and the per-agent kernels promise bit-identical repeats; the batched wide
learner adds with f32 atomics (~1e-6). Usage: tools/determinism_engines.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_mlp import _perturbed_group

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
P, M, B, S = 4096, (int(sys.argv[2]) if len(sys.argv) > 2 else 5), 64, 4
conf, grp = _perturbed_group(M, S=S, seed=7)
g = torch.Generator(device="cuda").manual_seed(8)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
nA = P * M
s, a = 1.5 * rn(nA, B, S), 2.5 * (2 * torch.rand(nA, B, 1, device="cuda", generator=g) - 1)
r, s2 = -rn(nA, B).abs() * 0.3, 1.5 * rn(nA, B, S)
sm = lambda x: x.view(P, M, *x.shape[1:]).transpose(0, 1).reshape(M, P * B, *x.shape[2:]).contiguous()  # set-major for learn_shared
engines = {
    "split (fsplit.hip)": lambda: grp.learn_set_split(s, a, r, s2, nA),
    "fused bf16 (fset.hip)": lambda: grp.learn_set_fused(s, a, r, s2, nA),
    "batched (wide.hip, reference widths)": lambda: grp.learn_shared(sm(s), sm(a), sm(r), sm(s2), nA),
}
for name, fn in engines.items():
    ref = fn().clone()
    worst, bad = 0.0, 0
    for _ in range(N):
        out = fn()
        d = ((out - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, d)
        bad += d > 1e-4 or d != d
    print(f"{name:40s}: worst deviation over {N} repeats {worst:.2e}; repeats off: {bad}")
# per-agent f32 kernel on a subset of agents per set (its gradients are per agent)
sub = 4096
gs = torch.empty(sub, grp.lay.theta_size, device="cuda")
big = None
try:
    from avddpg_amd import vec
    ag = vec.AgentGroup(sub, S, 1, conf, seed=3)
    ref = ag.learn(s[:sub], a[:sub], r[:sub], s2[:sub], 0).clone()
    worst, bad = 0.0, 0
    for _ in range(max(20, N // 10)):
        out = ag.learn(s[:sub], a[:sub], r[:sub], s2[:sub], 0)
        d = ((out - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, d)
        bad += d > 1e-6
    print(f"{'per-agent f32 (lean.hip), 4096 agents':40s}: worst deviation {worst:.2e}; repeats off: {bad}")
except Exception as e:  # noqa
    print("per-agent check skipped:", e)

# centralized framework (cen::learn_kernel_c: S = 4 L, A = L, widths x 1.2), 2048 models: the learn kernel, and the whole update
# (chunks of 256 on the caller's stream, their Adam + Polyak passes on a side stream) restarted from the same state every time
try:
    L = 5
    cg = vec.AgentGroup(2048, 4 * L, L, conf, seed=5, hidd_mult=1.2)
    cs, ca = 1.5 * rn(2048, B, 4 * L), 2.5 * (2 * torch.rand(2048, B, L, device="cuda", generator=g) - 1)
    cr, cs2 = -rn(2048, B).abs() * 0.3, 1.5 * rn(2048, B, 4 * L)
    ref = cg.learn(cs, ca, cr, cs2, 0).clone()
    worst, bad = 0.0, 0
    for _ in range(max(20, N // 10)):
        out = cg.learn(cs, ca, cr, cs2, 0)
        d = ((out - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, d)
        bad += d > 1e-6
    print(f"{'centralized f32 (cen.hip), 2048 models':40s}: worst deviation {worst:.2e}; repeats off: {bad}")
    state0 = [x.clone() for x in (cg.theta, cg.theta_t, cg.stats_t, cg.m, cg.v, cg.step)]
    scratch = torch.empty(2048, cg.lay.theta_size, device="cuda")
    ref, bad, reps = None, 0, max(20, N // 20)
    for _ in range(reps):
        for dst, src in zip((cg.theta, cg.theta_t, cg.stats_t, cg.m, cg.v, cg.step), state0):
            dst.copy_(src)
        cg.learn_update(cs, ca, cr, cs2, scratch)
        cg.learn_update(cs2, ca, cr, cs, scratch)
        got = torch.cat([cg.theta.flatten(), cg.theta_t.flatten(), cg.stats_t.flatten(), cg.m.flatten(), cg.v.flatten()])
        if ref is None:
            ref = got.clone()
        bad += not torch.equal(got, ref)
    print(f"{'centralized update, 8 chunks x 2 streams':40s}: {reps} restarts of two updates, bitwise different from the first: {bad}")
except Exception as e:  # noqa
    print("centralized check skipped:", repr(e))
