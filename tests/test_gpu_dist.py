"""Two ranks of the REAL multi-GPU interfrl path on one device: each rank is its own process with its own VecTrainer
(own platoons, own Philox streams), gradients meet in the all-reduce of avddpg_amd/dist.py. RCCL refuses two ranks on
one GPU, so the process group is gloo over CUDA tensors -- the code path above the collective is the one
`bench.py --gpus N --mode interfrl` runs over RCCL. What must hold (reference workers/trainer.py:121-131, 400-431): all
ranks start from the same weights and, every step being federated, hold bit-identical weight sets ever after."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shared, q, engine=None):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from avddpg_amd import config, trainer

        conf = config.Config(num_platoons=5 + rank, pl_size=3, buffer_size=128, fed_method="interfrl",
                             weighted_average_enabled=False)  # unequal shards on purpose
        vt = trainer.VecTrainer(conf, rng="device", group=dist.group.WORLD, auto_reset=True, seed=1 + rank,
                                shared_sets=shared, shared_engine=engine)
        th0 = vt.agents.theta.clone()
        vt.reset_episode()
        for _ in range(70):  # the strict gate opens with the 65th add: 6 federated updates
            vt.step()
        torch.cuda.synchronize()
        th = vt.agents.theta.view(-1, vt.M, vt.agents.lay.theta_size) if not shared else vt.agents.theta[None]
        q.put((rank, vt.total_platoons, th0[0].cpu().numpy(), th.cpu().numpy(), vt.env.x.cpu().numpy()[:5],
               int(vt.agents.step[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shared", [True, False, "fused"])
def test_two_rank_interfrl_weight_sets_stay_identical(shared):
    """shared = "fused": shared sets with the fused bf16 set learner (csrc/fset.hip): each rank's mean gradient is scaled to a
    sum over its platoons, all-reduced, divided by the global platoon count -- deterministic, so the ranks stay bit-identical."""
    engine = "fused" if shared == "fused" else None
    shared = bool(shared)
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shared, q, engine)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, tot0, init0, th_a, x_a, st_a), (_, tot1, init1, th_b, x_b, st_b) = res
    assert tot0 == tot1 == 11.0 and st_a == st_b == 6
    assert np.array_equal(init0, init1)                # same initial weights on both ranks (rank-invariant seed + broadcast)
    assert not np.array_equal(x_a, x_b)                # different platoons / streams
    assert not np.array_equal(th_a[0, 0], init0)       # the sets did learn
    # every platoon's copy of vehicle m's set, on both ranks, is the same bits
    for m in range(th_a.shape[1]):
        ref = th_a[0, m]
        assert all(np.array_equal(ref, th_a[p, m]) for p in range(th_a.shape[0]))
        assert all(np.array_equal(ref, th_b[p, m]) for p in range(th_b.shape[0]))


def test_bench_gpus2_self_spawned_ranks_run_both_workloads_and_print_one_line():
    """`python bench.py --gpus 2` with no launcher: the script starts its two ranks itself (children, before any GPU call),
    runs nofrl (replicas) and interfrl (one all-reduce of the gradient slab per step) and rank 0 prints ONE JSON line.
    On a 1-GPU box: --single-device + gloo (RCCL refuses two ranks per device); on an N-GPU node the same command
    without those two flags runs over RCCL."""
    import json
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--platoons", "96",
           "--buffer-size", "512", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak"
    modes = {out["config"]["mode"]: out} | out["also_measured"]
    assert set(modes) == {"nofrl", "interfrl"}
    for m, r in modes.items():
        assert r["value"] > 0 and abs(r["value"] - 2 * 96 * 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]
        assert r["roofline"]["bound"] in ("hbm", "mfma") and r["roofline"]["achieved"] > 0
    assert modes["interfrl"]["collective_backend"] == "gloo" and "all-reduce" in modes["interfrl"]["config"]["parallelism"]
    assert "no data-path collective" in modes["nofrl"]["config"]["parallelism"]
