"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch

from avddpg_amd import config
from oracle import platoon


def need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda")


def conf_and_ep(**kw):
    """Matching (product Config, oracle EnvParams) pair."""
    conf = config.Config(**kw)
    ep = platoon.EnvParams()
    for k, v in kw.items():
        if hasattr(ep, k):
            setattr(ep, k, v)
    return conf, ep


def golden_case_kwargs(key):
    kw = {}
    if "ModelA" in key or key.endswith("_A"):
        kw["model"] = "ModelA"
    if "exact" in key:
        kw["method"] = "exact"
    if key.startswith("terminal_off"):
        kw["can_terminate"] = False
    if key.startswith("nondegenerate"):
        kw.update(dyn_coeff=0.25, pl_leader_tau=0.15, timegap=0.8, sample_rate=0.05)
        if key.endswith("_B"):
            kw.update(method="exact", re_scalar=2.0)
    if key.startswith("centralized"):
        kw["framework"] = "centralized"
    return kw


def t(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype).cuda()
