"""bench.py as the driver starts it for N > 1: self-spawned ranks, one all-reduce(sum) of the set gradients per step
(`workers/trainer.py:400-431` averaged over the platoon shards). On a one-GPU box the two ranks share the device over gloo
(RCCL refuses two ranks per device); with two or more GPUs visible the same command runs over RCCL ("nccl")."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "interfrl", "--platoons", "256", "--buffer-size", "4096",
           "--steps", "6", "--warmup", "2", "--no-cpu-baseline"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_over_gloo():
    need_gpu()
    out = _bench(["--backend", "gloo", "--single-device"])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak" and out["config"]["platoons_per_gpu"] == 256
    assert out["collective_backend"] == "gloo" and "all-reduce" in out["config"]["parallelism"]


def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    need_gpu()
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks per device (covered over gloo above and in tests/test_dist_cpu.py)")
    out = _bench(["--backend", "nccl"])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["rccl_ranks"] == 2 and out["collective_backend"] == "nccl"
