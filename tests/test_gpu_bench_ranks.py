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


def test_default_one_gpu_run_carries_every_single_gpu_baseline_config():
    """The driver's command (`python bench.py --gpus 1 --steps K --warmup W`) times configs[1] in both federated modes AND
    configs[2] (4096 x 10, both modes, replay capacity 100000) AND configs[4] (hidden 1024) in ONE line: `value` stays
    configs[1] interfrl, the others sit under also_measured with their own roofline. Full sizes (82 + 164 GB rings): about a minute."""
    need_gpu()
    if torch.cuda.get_device_properties(0).total_memory < 250 * 2**30:
        pytest.skip("needs the 288 GB of an MI355X (4096 x 10 at replay capacity 100000)")
    sys.path.insert(0, ROOT)
    import bench

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--prewarm-seconds", "0.2",
           "--no-cpu-baseline"]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["primary_mode"] == "interfrl" and out["config"]["pl_size"] == 5 and out["value"] > 0 and out["n_gpus"] == 1
    also = out["also_measured"]
    assert set(also) == {"nofrl"} | set(bench.EXTRA_CONFIGS)
    for k, name in bench.EXTRA_CONFIGS.items():
        r = also[k]
        assert r["value"] > 0 and r["steps"] == 4 and r["config"]["baseline_config"] == name
        assert r["roofline"]["bound"] in ("hbm", "mfma") and 0 < r["roofline"]["frac"] < 1
    assert also["config3_interfrl"]["config"]["pl_size"] == 10 and also["config3_nofrl"]["config"]["mode"] == "nofrl"
    assert "hidden=1024" in also["config5"]["config"]["workload"] and also["config5"]["roofline"]["unit"] == "TFLOP/s"
