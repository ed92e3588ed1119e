import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_mlp import _perturbed_group, _nets
from oracle import mlp as omlp
def bf(x):
    return torch.tensor(np.asarray(x, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()
S, P, (H1, H2, Ha) = 4, int(os.environ.get('DBG_P', '8')), (1024, 1024, 48)
n_sets, B = 2, 64
conf, grp = _perturbed_group(n_sets, S=S, seed=61, actor_layer1_size=H1, actor_layer2_size=H2, critic_layer1_size=H1, critic_layer2_size=H2, critic_act_layer_size=Ha)
rs = np.random.RandomState(62); rows = P * B
s = rs.normal(0, 1.5, size=(n_sets, rows, S)).astype(np.float32)
da_gpu = {d: np.fromfile(f"/tmp/da{d}.bin", np.float32).reshape(3, 2, -1) for d in "01"} if os.path.exists("/tmp/da1.bin") else None
for k in range(n_sets):
    a_w, c_w, _, _ = _nets(grp, k, np.float64)
    mu = omlp.actor_forward(a_w, s[k], 2.5)
    if da_gpu: mu = da_gpu['1'][2, k][:rows, None].astype(np.float64)  # the GPU's own mu
    Ws, bs, Wa, ba, gs, bes, mms, mvs, ga, bea, mma, mva, W2, b2, g3, be3, mm3, mv3, W3, b3 = c_w
    q, (s_, a_, ps, pa, c, p2, y2) = omlp.critic_forward(c_w, s[k], mu, cache=True)
    N = rows
    _, da = omlp.critic_backward(c_w, (s_, a_, ps, pa, c, p2, y2), np.full((N, 1), -1.0 / N), need_params=False)
    ia, _ = omlp._bn_coeffs(ga, bea, mma, mva); i3, _ = omlp._bn_coeffs(g3, be3, mm3, mv3)
    cf = (i3 * W3[:, 0]); mask = (p2 > 0).astype(np.float64); pos = (pa > 0).astype(np.float64)
    W2a = W2[H1:, :]  # [Ha][H2]
    drow = -1.0 / N
    E_exact = mask @ (cf[None, :] * (ia[:, None] * W2a)).T     # [N][Ha]
    da_exact = drow * (pos * Wa[0][None, :] * E_exact).sum(1)
    E_dual = (mask * bf(cf)[None, :]) @ bf(ia[:, None] * W2a).T
    da_dual = drow * (pos * Wa[0][None, :] * E_dual).sum(1)
    E_delta = mask @ bf(cf[None, :] * W2a).T
    da_delta = drow * (pos * (ia * Wa[0])[None, :] * E_delta).sum(1)
    print("set", k, "oracle da[:4]", da[:4, 0], "formula", da_exact[:4])
    print("   emul dual - exact", (da_dual - da_exact)[:6], "emul delta - exact", (da_delta - da_exact)[:6])
    if da_gpu:
        sc = np.abs(da_exact).max()
        for nm, d in (("dual", da_gpu["1"][1, k]), ("delta", da_gpu["0"][1, k])):
            e = np.abs(d - da_exact) / sc
            if nm == "dual":
                dd = np.abs(da_gpu["1"][1, k] - da_gpu["0"][1, k]) / sc
                idx = np.argsort(-dd)[:8]
                print("    rows where the paths differ most", idx.tolist(), "diff", dd[idx].round(3).tolist(), "dual err", (np.abs(da_gpu["1"][1, k] - da_exact) / sc)[idx].round(3).tolist(), "delta err", (np.abs(da_gpu["0"][1, k] - da_exact) / sc)[idx].round(3).tolist())
            print("   ", nm, "max err / max|da|", e.max(), "rows > 2 %:", int((e > 0.02).sum()), "worst rows", np.argsort(-e)[:6].tolist(), np.sort(-e)[:6].round(3).tolist())
        # what the worst dual row looks like: margin of p1(mu) to zero
        w = int(np.argmax(np.abs(da_gpu["1"][1, k] - da_exact)))
        p1mu = (mu[w:w+1] @ Wa + ba)[0]
        print("    worst dual row", w, "min |p1(mu)| over features", np.abs(p1mu).min(), "a(mu)", mu[w, 0], "contrib of that feature", (Wa[0] * E_exact[w] * drow)[int(np.argmin(np.abs(p1mu)))], "da_exact", da_exact[w], "gpu dual", da_gpu["1"][1, k][w], "gpu delta", da_gpu["0"][1, k][w])
        print("   gpu dual - exact ", (da_gpu["1"][1, k] - da_exact)[:6], "gpu delta - exact ", (da_gpu["0"][1, k] - da_exact)[:6])
        # per-feature attribution of the gpu dual offset: regress (gpu_dual - exact)/drow on pos*Wa*E_exact columns
        X = pos * Wa[0][None, :] * E_exact * drow
        coef, *_ = np.linalg.lstsq(X, da_gpu["1"][1, k] - da_exact, rcond=None)
        print("   lstsq coef per feature (dual): ", np.round(coef, 3).tolist())
        coef0, *_ = np.linalg.lstsq(X, da_gpu["0"][1, k] - da_exact, rcond=None)
        print("   lstsq coef per feature (delta):", np.round(coef0, 3).tolist())
