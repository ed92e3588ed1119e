"""HEAD_BOTH against the two-launch form (AVD_FSPLIT_TWO_HEADS) and the exact-f32 engine on the inputs of
tests/test_gpu_fsplit.py::test_split_set_learner_matches_oracle_at_the_f32_tolerance[4-70-5]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avddpg_amd import vec
from tests.gpu_util import t
from tests.test_gpu_mlp import _perturbed_group
from tests.test_gpu_fset import NAMES, _batch
P, M = int(os.environ.get("P", 70)), 5
conf, grp = _perturbed_group(M, S=4, seed=61)
rs = np.random.RandomState(62)
n = P * M
s, a, r, s2 = _batch(rs, n, 4)
per = grp.learn(t(s), t(a), t(r), t(s2), M)
avg = vec.fed_mean(per, P, M, method=conf.interfrl)
def report(tag, g, ref):
    print(tag)
    for k in range(M):
        cs, as_ = grp.grads_as_lists(g[k]); ce, ae = grp.grads_as_lists(ref[k])
        bad = [(nm, float(np.abs(x - z).max() / (np.abs(z).max() + 1e-30))) for nm, x, z in zip(NAMES, cs + as_, ce + ae)]
        print("  set", k, " ".join(f"{nm}:{e:.1e}" for nm, e in bad if e > 2e-5) or "ok", " max err %.1e" % max(e for _, e in bad))
g1 = grp.learn_set_split(t(s), t(a), t(r), t(s2), n).clone()
os.environ["AVD_FSPLIT_TWO_HEADS"] = "1"
g2 = grp.learn_set_split(t(s), t(a), t(r), t(s2), n).clone()
report("HEAD_BOTH vs f32", g1, avg)
report("two heads vs f32", g2, avg)
report("HEAD_BOTH vs two heads", g1, g2)


# ---- per-row dmu of the two forms (workspace offsets restated from make_plan in csrc/fsplit.hip) ----
def dmu_offset(n_agents, n_sets):
    o = 0
    def take(b):
        nonlocal o
        at = o
        o += (b + 255) // 256 * 256
        return at
    H2, VEC, NGT_MAX = 128, 264, 10
    for i in range(4):
        K, KP = (304, 320) if i & 1 else (256, 256)
        take(2 * n_sets * H2 * K); take(2 * n_sets * H2 * K); take(4 * n_sets * VEC)
        take(16 * n_sets * NGT_MAX * 64); take(16 * n_sets * NGT_MAX * 64)
        if i < 2:
            take(2 * n_sets * KP * H2); take(2 * n_sets * KP * H2)
    take(4 * n_sets * 48)
    rows = n_agents * 64
    take(4 * rows); take(4 * rows); take(4 * rows)
    return take(4 * rows), rows

for PP in (70, 4096):
    n = PP * M
    rs = np.random.RandomState(62)
    s, a, r, s2 = _batch(rs, n, 4)
    off, rows = dmu_offset(n, M)
    out = {}
    for two in (False, True):
        if two: os.environ["AVD_FSPLIT_TWO_HEADS"] = "1"
        else: os.environ.pop("AVD_FSPLIT_TWO_HEADS", None)
        grp.learn_set_split(t(s), t(a), t(r), t(s2), n)
        torch.cuda.synchronize()
        ws = grp._fsplit_ws
        out[two] = ws.view(torch.uint8)[off:off + 4 * rows].view(torch.float32).clone().cpu().numpy()
    d = np.abs(out[False] - out[True]); sc = np.abs(out[True]).max()
    print(f"P={PP}: rows {rows}  dmu max {sc:.3e}  rows differing by > 1e-5 max: {(d > 1e-5 * sc).sum()}  > 1e-3 max: {(d > 1e-3 * sc).sum()}  median diff/max {np.median(d) / sc:.1e}  max diff/max {d.max() / sc:.1e}")
    bad = np.nonzero(d > 1e-5 * sc)[0]
    for b in bad[:10]:
        from oracle import mlp as omlp
        from tests.test_gpu_mlp import _nets
        ag, rw = b // 64, b % 64
        an, cn, _, _ = _nets(grp, ag % M, np.float64)
        srow = s[ag][rw:rw + 1].astype(np.float64)
        mu = omlp.actor_forward(an, srow, 2.5)
        q, (s_, a_, ps, pa, c, p2, y2) = omlp.critic_forward(cn, srow, mu, cache=True)
        Wa, ba, W2, b2 = cn[2], cn[3], cn[12], cn[13]
        za = (mu @ Wa + ba)[0]; z2 = (c @ W2 + b2)[0]
        print("   mu", mu.ravel(), " min |za|", np.abs(za).min(), "at", np.abs(za).argmin(), " min |z2|", np.abs(z2).min(), "at", np.abs(z2).argmin(), " scales", np.abs(za).max(), np.abs(z2).max())
        print("   row", b, "agent", b // 64, "set", (b // 64) % M, "platoon", b // 64 // M, "row in tile", b % 64, "both", out[False][b], "two", out[True][b])
