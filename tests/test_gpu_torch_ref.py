"""The plain-PyTorch reference of the set learners (tools/torch_set_learn.py: float32 / float64 autograd on the slabs) pinned against
the float64 NumPy oracle, and the split-operand engine against IT -- the torch fp32 reference for the floating-point kernels, usable
at widths where no exact-f32 HIP engine exists (hidden 1024: tests/test_gpu_wide.py, tools/train_curves.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import mlp as omlp  # noqa: E402
from tests.gpu_util import need_gpu, t  # noqa: E402
from tests.test_gpu_fset import NAMES, _batch  # noqa: E402
from tests.test_gpu_fsplit import SPLIT_TOL  # noqa: E402
from tests.test_gpu_mlp import _nets, _perturbed_group, _relerr  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,P,M", [(4, 6, 2), (3, 5, 3)])
def test_torch_reference_equals_the_float64_oracle_and_the_split_engine_tracks_it(S, P, M):
    need_gpu()
    import torch_set_learn as tsl

    conf, grp = _perturbed_group(M, S=S, seed=71)
    n = P * M
    s, a, r, s2 = _batch(np.random.RandomState(72), n, S)
    ts, ta, tr, ts2 = t(s), t(a), t(r), t(s2)
    g64 = tsl.learn_sets(grp, ts, ta, tr, ts2, n, dtype=torch.float64)
    g32 = tsl.learn_sets(grp, ts, ta, tr, ts2, n, dtype=torch.float32)
    gsp = grp.learn_set_split(ts, ta, tr, ts2, n)
    for k in range(M):
        sel = np.arange(k, n, M)
        cat = lambda x: x[sel].reshape(len(sel) * 64, *x.shape[2:])
        cg, ag, _ = omlp.learn((cat(s), cat(a), cat(r)[:, None], cat(s2)), *_nets(grp, k, np.float64))
        for name, ref, x64, x32, xsp in zip(NAMES, cg + ag, sum(grp.grads_as_lists(g64[k]), []), sum(grp.grads_as_lists(g32[k]), []),
                                            sum(grp.grads_as_lists(gsp[k]), [])):
            assert _relerr(x64, ref) <= 2e-7, (k, name, _relerr(x64, ref))      # (float64 autograd, rounded to the float32 slab)
            assert _relerr(x32, ref) <= 1e-4, (k, name, _relerr(x32, ref))      # the exact-f32 kernels' tolerance
            assert _relerr(xsp, x64) <= max(SPLIT_TOL, 4 * _relerr(x32, ref)), (k, name, _relerr(xsp, x64))
