"""dual vs delta path of wide.hip on the same inputs (diagnostic library, AVD_WIDE_DUAL=0/1 in separate processes)."""
import os, sys, subprocess
import numpy as np
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
    from tests.test_gpu_mlp import _perturbed_group
    from tests.gpu_util import t
    S, P, (H1, H2, Ha) = 4, int(os.environ.get('DBG_P', '8')), (1024, 1024, 48)
    n_sets, B = 2, 64
    conf, grp = _perturbed_group(n_sets, S=S, seed=61, actor_layer1_size=H1, actor_layer2_size=H2, critic_layer1_size=H1,
                                 critic_layer2_size=H2, critic_act_layer_size=Ha)
    rs = np.random.RandomState(62)
    rows = P * B
    s = rs.normal(0, 1.5, size=(n_sets, rows, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n_sets, rows, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n_sets, rows))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n_sets, rows, S)).astype(np.float32)
    losses = torch.zeros(n_sets, 2, device="cuda")
    g = grp.learn_shared(t(s), t(a), t(r), t(s2), n_sets * P, losses=losses)
    torch.cuda.synchronize()
    np.save(sys.argv[1], g.cpu().numpy())
    print(sys.argv[1], "losses", losses.cpu().numpy().tolist())
    from oracle import mlp as omlp
    from tests.test_gpu_mlp import _nets, _relerr
    for k in range(n_sets):
        cg, ag, aux = omlp.learn((s[k], a[k], r[k][:, None], s2[k]), *_nets(grp, k, np.float64))
        gcg, gag = grp.grads_as_lists(g[k])
        names = ["aW1", "ab1", "ag1", "abe1", "aW2", "ab2", "ag2", "abe2", "aW3", "ab3"]
        print(" set", k, "oracle aux", aux if not hasattr(aux, "keys") else {kk: (float(np.mean(v)) if np.ndim(v) else float(v)) for kk, v in aux.items()})
        print("   ", " ".join(f"{nm}:{_relerr(got, ref):.3f}" for nm, got, ref in zip(names, gag, ag)))
        print("    ab3 got/ref", np.ravel(gag[-1]), np.ravel(ag[-1]), "aW3 ratio of norms", np.linalg.norm(gag[-2]) / np.linalg.norm(ag[-2]),
              "cos", float(np.sum(np.ravel(gag[-2]) * np.ravel(ag[-2])) / np.linalg.norm(gag[-2]) / np.linalg.norm(ag[-2])))
else:
    root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
    env = dict(os.environ, AVDDPG_HIP_LIB=root + "/avddpg_amd/lib/libavddpg_hip_diag.so")
    for d in ("0", "1"):
        subprocess.run([sys.executable, __file__, f"/tmp/g{d}.npy"], env=dict(env, AVD_WIDE_DUAL=d, AVD_WIDE_DUMP=f"/tmp/da{d}.bin"), check=True)
    g0, g1 = np.load("/tmp/g0.npy"), np.load("/tmp/g1.npy")
    d0, d1 = np.fromfile("/tmp/da0.bin", np.float32).reshape(3, 2, -1), np.fromfile("/tmp/da1.bin", np.float32).reshape(3, 2, -1)
    for k in range(2):
        print("set", k, "q(s,mu) max diff", np.abs(d0[0, k] - d1[0, k]).max(), "da: |delta|max", np.abs(d0[1, k]).max(), "ratio dual/delta median",
              np.median(d1[1, k] / d0[1, k]), "quantiles", np.quantile(d1[1, k] / d0[1, k], [0.05, 0.25, 0.75, 0.95]).tolist())
        bad = np.nonzero(np.abs(d1[1, k] - d0[1, k]) > 0.02 * np.abs(d0[1, k]).max())[0]
        print("   max |dual - delta| / max|delta|", np.abs(d1[1, k] - d0[1, k]).max() / np.abs(d0[1, k]).max())
        print("   bad rows", len(bad), "of", d0.shape[2], "first", bad[:40].tolist(), "mod 128 hist", np.bincount(bad % 128 // 16, minlength=8).tolist(), "tile hist", np.bincount(bad // 128, minlength=4).tolist())
    print("shape", g0.shape, "max|g0|", np.abs(g0).max(), "max diff", np.abs(g0 - g1).max())
    asz = None
    d = np.abs(g0 - g1)
    for k in range(g0.shape[0]):
        idx = np.argsort(-d[k])[:5]
        print("set", k, "worst idx", idx.tolist(), d[k][idx].tolist(), g0[k][idx].tolist(), g1[k][idx].tolist())
        # relative error per 1/16 of the slab
        n = g0.shape[1]
        for i in range(16):
            sl = slice(i * n // 16, (i + 1) * n // 16)
            print("  part", i, "max|ref|", float(np.abs(g0[k][sl]).max()), "max diff", float(d[k][sl].max()))
