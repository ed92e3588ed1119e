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
