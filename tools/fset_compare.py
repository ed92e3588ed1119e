import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from avddpg_amd import config, vec
from tests.test_gpu_mlp import _perturbed_group
from tests.gpu_util import t
for P, M in ((8, 3), (1, 1), (70, 5), (512, 5)):
    conf, grp = _perturbed_group(M, S=4, seed=71)
    rs = np.random.RandomState(72)
    n = P * M; B = 64
    s = rs.normal(0, 1.5, size=(n, B, 4)).astype(np.float32); a = rs.uniform(-2.5, 2.5, size=(n, B, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n, B))).astype(np.float32); s2 = rs.normal(0, 1.5, size=(n, B, 4)).astype(np.float32)
    per_agent = grp.learn(t(s), t(a), t(r), t(s2), M)
    avg = vec.fed_mean(per_agent, P, M, method=conf.interfrl)
    sm = lambda x: t(np.ascontiguousarray(x.reshape(P, M, *x.shape[1:]).swapaxes(0, 1)).reshape(M, P * B, *x.shape[2:]))
    wide = grp.learn_shared(sm(s), sm(a), sm(r), sm(s2), n)
    fused = grp.learn_set_fused(t(s), t(a), t(r), t(s2), n)
    lay = grp.lay
    for name, lo, hi in (("actor", 0, lay.actor_size), ("critic", lay.actor_size, lay.theta_size)):
        sc = avg[:, lo:hi].abs().max().item()
        print(P, M, name, "fused-f32 %.2e  wide-f32 %.2e  fused-wide %.2e" % ((fused[:, lo:hi] - avg[:, lo:hi]).abs().max().item() / sc,
              (wide[:, lo:hi] - avg[:, lo:hi]).abs().max().item() / sc, (fused[:, lo:hi] - wide[:, lo:hi]).abs().max().item() / sc))
    # per tensor
    if P == 8:
        names = ["aW1","ab1","ag1","abe1","aW2","ab2","ag2","abe2","aW3","ab3"]
        ga, _ = None, None
        cgf, agf = grp.grads_as_lists(fused[0]); cgw, agw = grp.grads_as_lists(wide[0]); cga, aga = grp.grads_as_lists(avg[0])
        for nm, f_, w_, a_ in zip(names, agf, agw, aga):
            sc = np.abs(a_).max() + 1e-30
            print("   ", nm, "fused %.2e wide %.2e" % (np.abs(f_ - a_).max() / sc, np.abs(w_ - a_).max() / sc))
