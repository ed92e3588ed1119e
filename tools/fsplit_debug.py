import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avddpg_amd import vec
from tests.gpu_util import t
from tests.test_gpu_mlp import _perturbed_group
from tests.test_gpu_fset import NAMES, _batch
P, M = 51, 5
conf, grp = _perturbed_group(M, S=4, seed=61)
rs = np.random.RandomState(62)
n = P * M
s, a, r, s2 = _batch(rs, n, 4)
per = grp.learn(t(s), t(a), t(r), t(s2), M)
avg = vec.fed_mean(per, P, M, method=conf.interfrl)
def report(tag, g):
    print(tag)
    for k in range(M):
        cs, as_ = grp.grads_as_lists(g[k]); ce, ae = grp.grads_as_lists(avg[k])
        bad = [(nm, float(np.abs(x - z).max() / (np.abs(z).max() + 1e-30))) for nm, x, z in zip(NAMES, cs + as_, ce + ae)]
        print("  set", k, " ".join(f"{nm}:{e:.1e}" for nm, e in bad if e > 5e-5) or "ok", " max err %.1e" % max(e for _, e in bad))
report("full call, default J", grp.learn_set_split(t(s), t(a), t(r), t(s2), n).clone())
# mean of single-platoon calls
acc = torch.zeros_like(avg)
for p in range(P):
    sl = lambda x: t(x[p * M:(p + 1) * M])
    acc += grp.learn_set_split(sl(s), sl(a), sl(r), sl(s2), M)
report("mean of 51 single-platoon calls", acc / P)
# single platoon 0 alone vs per-agent kernel on platoon 0
g0 = grp.learn_set_split(t(s[:M]), t(a[:M]), t(r[:M]), t(s2[:M]), M)
for k in range(M):
    cs, as_ = grp.grads_as_lists(g0[k]); ce, ae = grp.grads_as_lists(per[k])
    bad = [(nm, float(np.abs(x - z).max() / (np.abs(z).max() + 1e-30))) for nm, x, z in zip(NAMES, cs + as_, ce + ae)]
    print("  platoon 0 alone, set", k, " max err %.1e" % max(e for _, e in bad), " ".join(f"{nm}:{e:.1e}" for nm, e in bad if e > 5e-5))
for J in (8, 16, 32, 33, 40):
    os.environ["AVD_FSPLIT_J"] = str(J)
    grp._fsplit_ws = None
    report(f"full call, J={J}", grp.learn_set_split(t(s), t(a), t(r), t(s2), n).clone())
