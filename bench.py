#!/usr/bin/env python3
"""Headline benchmark: env-steps/s + DDPG updates/s of the avddpg hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic platoons on each GPU:
  actor forward -> OU noise -> clip -> leader exog -> platoon step -> replay add -> replay sample
  -> Trainer.learn (5 forwards, 2 backwards, B=64 per agent) -> Adam x2 + Polyak -> episode reset.
Workload (BASELINE.json configs[1]): 4096 platoons x 5 vehicles per GPU, reference `nofrl` DDPG = one
independent actor/critic/target/Adam/replay set per (platoon, vehicle) = 20480 agents, f32, replay at its
steady state (capacity 100000 rows per agent, full).  `--mode interfrl` runs configs[3]'s federated variant
(one weight set per vehicle index, gradients averaged over platoons and all-reduced across ranks with RCCL).

Launch: python bench.py --gpus N --steps K --warmup W. N > 1: one rank per GPU, either under torch.distributed.run
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when WORLD_SIZE is not set, started BY this script as N child
processes before anything touches the GPU (the parent only waits and never imports torch). Rank 0 prints ONE JSON line.

On one GPU, without flags that select a workload, the same run then also times the other single-GPU BASELINE configs --
configs[2] (4096 x 10, interfrl and nofrl at replay capacity 100000) and configs[4] (hidden 1024, bf16 batched engine) -- each
with its own prewarm, warm-up and timed region of the same --steps (capped at 300), under also_measured.{config3_interfrl,
config3_nofrl, config5}; `value` stays configs[1] interfrl (--no-extra-configs skips them).

Without --mode the line carries BOTH workloads of the 4096 x 5 shape, measured one after the other in the same run:
the primary one (PRIMARY_MODE) as `value`, the other under `also_measured`:
  * nofrl    -- the reference's default `fed_method` (src/config.py:25): 20480 independent agents, HBM-bound, no
                data-path collective (replicas only across GPUs);
  * interfrl -- BASELINE configs[3]'s per-GPU shape (workers/trainer.py:400-431): gradients averaged over the platoons
                of ALL ranks through one RCCL all-reduce per step (`rccl_ranks` = N).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak (= f32 vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline figure is 2:1 sparse)
LEARN_FLOP_PER_SAMPLE = 0.751e6  # 5 forwards + 2 backwards per replay sample
ADAM_BYTES_PER_AGENT = 2.47e6  # r/w of W, W_target, m, v per agent-update


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the clocks to settle (the first ~20 steps after an idle chip run 5 % slow; r03: 20 / 3 steps read
    # 1.65 M where 100+ steps read 1.74 M), short enough for the default run to finish in about a minute
    # >= 2000 timed steps after 200 warm-up. (The first ~0.5 s after an idle chip run 3-5 % slow: 20 + 200 steps of
    # the 2.3 ms interfrl step read 1.64-1.74 M where 2000 sustained read 1.69-1.79 M on the same box.)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--platoons", type=int, default=4096, help="platoons per GPU")
    ap.add_argument("--pl-size", type=int, default=5)
    ap.add_argument("--buffer-size", type=int, default=100000)
    ap.add_argument("--mode", choices=["nofrl", "interfrl", "intrafrl"], default=None,
                    help="measure this workload only (default: both, PRIMARY_MODE as `value`, the other under `also_measured`)")
    ap.add_argument("--no-secondary", action="store_true", help="default run: skip the non-primary workload")
    ap.add_argument("--engine", choices=["per_agent", "batched", "fused", "fused3"], default=None,
                    help="interfrl: per_agent = f32 LDS-resident learn kernel per agent + federated sum; batched = one "
                         "bf16 MFMA GEMM chain per weight set over all its rows (default where per_agent does not exist); "
                         "fused = persistent set learner, bf16 operands (fset.hip); fused3 = the same with every operand an "
                         "fp16 hi+lo pair, f32-class results (fsplit.hip)")
    ap.add_argument("--hidden", type=int, default=None,
                    help="actor/critic layer1 = layer2 size (BASELINE config 5: 1024; needs --mode interfrl)")
    ap.add_argument("--framework", choices=["decentralized", "centralized"], default="decentralized",
                    help="centralized: one model per platoon with S = 4L, A = L, widths x1.2 (SURVEY 8 f-3; cen.hip / general learn kernel)")
    ap.add_argument("--chunks", type=int, default=1,
                    help="nofrl: agent slices for the learn || Adam+Polyak two-stream pipeline (1 = serial). intrafrl: platoon chunks of the "
                         "learn || mean + Adam + Polyak pipeline (--intra-chunks)")
    ap.add_argument("--intra-chunks", type=int, default=16,
                    help="intrafrl: platoon chunks of the learn || mean + Adam + Polyak two-stream pipeline (1 = learn, then one pass)")
    ap.add_argument("--directional", action="store_true",
                    help="intrafrl: intra_directional_averaging -- the lead vehicle of every platoon takes no federated step (the reference "
                         "CLI's default, src/cmd/api.py:81; the Config default is off, src/config.py:37)")
    ap.add_argument("--no-fused", action="store_true",
                    help="nofrl: run learn and Adam+Polyak as two kernels instead of the fused avd_learn_update_f32")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--single-device", action="store_true",
                    help="plumbing test: every rank uses GPU 0 (a 1-GPU box cannot host one rank per GPU)")
    ap.add_argument("--overlap", action="store_true",
                    help="interfrl, N > 1: the critic block's all-reduce on a side stream under the actor phase of the learn call + the actor "
                         "block's on the main stream, instead of ONE all-reduce of the whole [M, theta] slab between learn and Adam (the "
                         "default: on a one-rank RCCL communicator the overlapped form measured slower, DESIGN.md section 6)")
    ap.add_argument("--no-overlap", action="store_true", help="(the default since r05; accepted for older command lines)")
    ap.add_argument("--no-collective-ab", action="store_true",
                    help="interfrl, N > 1: do not also time the OTHER collective form (by default the line carries both under "
                         "collective.forms so that one multi-GPU run attributes its own communication cost)")
    ap.add_argument("--init-timeout", type=float, default=600.0,
                    help="N > 1: seconds the process-group rendezvous + first collective may take before the rank gives up (non-zero exit). "
                         "Generous on purpose: on a fresh box the ranks' first `import torch` / HIP initialisation can finish minutes apart, "
                         "and the ranks that are ready wait here for the last one")
    ap.add_argument("--one-rank-rccl", action="store_true",
                    help="N = 1 only: run interfrl through a real RCCL communicator of ONE rank (every collective of the N > 1 path is "
                         "issued -- what a collective costs on the device with no wire behind it; not the default measurement)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5,
                    help="untimed steps of the SAME workload run before the --warmup steps until this much wall time has passed (0 = none): "
                         "a chip coming out of idle (trainer construction, ring fill) runs its first ~0.5 s 3-5 %% slow (clocks and power "
                         "state), so a short run -- --steps 20 --warmup 5 is 55 ms -- would time the ramp, not the workload; the line "
                         "records what was added (prewarm)")
    ap.add_argument("--weighted", action="store_true",
                    help="interfrl: weighted federated averaging (weighted_average_enabled, the Config default src/config.py:28; "
                         "workers/trainer.py:385-398) with the weights computed on the device from a ring of closed-episode rewards "
                         "(avd_fed_history_push_f32 + avd_fed_weights_f32: two small launches per step)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="default 1-GPU run: do not also time BASELINE configs[2] (4096 x 10) and configs[4] (hidden 1024) after the two "
                         "4096 x 5 workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time N independent single-thread copies of the CPU loop on all host cores (side figure, opt-in)")
    ap.add_argument("--master-port", type=int, default=0, help="self-spawned ranks: rendezvous port (0 = pick a free one)")
    ap.add_argument("--allow-diagnostics", action="store_true",
                    help="run although AVD_* / AVDDPG_HIP_LIB environment overrides are set or the loaded library is a diagnostic "
                         "build (the line then lists them under env_overrides / diagnostic_library); without it bench.py refuses")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def cpu_baseline(seconds, pl_size, mode="nofrl"):
    """The oracle's reference-shaped Python loop (oracle/trainer.py, kind "port": TensorFlow is not installable,
    so the NN arithmetic is NumPy float32) on ONE host thread -- the reference's own setting (src/rand.py:14-15) -- in the
    same federated mode as the line's primary workload (the reference's per-agent cost is the same in both: every agent
    learns on its own batch; interfrl adds the server's averaging)."""
    import numpy as np
    from threadpoolctl import threadpool_limits

    from oracle import platoon, trainer

    P = 2
    with threadpool_limits(limits=1):
        tr = trainer.RefTrainer(platoon.EnvParams(), P, pl_size, seed=1, buffer_size=4096,
                                fed_method="interfrl" if mode == "interfrl" else "normal")
        tr.reset_episode()
        for _ in range(65):  # replay warm-up: updates start at the 65th add
            if tr.step():
                tr.reset_episode()
        n, t0, u0 = 0, time.perf_counter(), tr.updates
        while time.perf_counter() - t0 < seconds:
            if tr.step():
                tr.reset_episode()
            n += 1
        dt = time.perf_counter() - t0
    return {"value": P * n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "updates_per_s": (tr.updates - u0) / dt,
            "sample": f"{P} platoons x {pl_size} vehicles (={P * pl_size} agents), {mode}, {n} training steps after a "
                      f"65-step replay warm-up, {dt:.1f} s on 1 thread; the per-platoon cost of the reference loop "
                      "does not depend on the number of platoons",
            "see_also": "profiles/r05_cpu_baseline_config1.txt (tools/cpu_baseline_config1.py): SURVEY 8(d)'s line -- BASELINE configs[0], 1 platoon x 3 "
                        "vehicles, 5000 steps on one thread: 362 env-steps/s; 256 independent single-thread copies on all host cores: 7.2 k"}


def _cpu_worker(seconds, pl_size, q, mode="nofrl"):
    r = cpu_baseline(seconds, pl_size, mode)
    q.put((r["value"], r["updates_per_s"]))


def cpu_baseline_all_cores(seconds, pl_size, max_workers=None, mode="nofrl"):
    """platoons are independent, so the host's whole-socket figure is N independent single-thread copies of
    the reference-shaped loop, N = the cores this process may run on. Reported beside the 1-thread figure, never as it.
    Opt-in (--cpu-all-cores). One overall deadline; stragglers are terminated together and the sum over the workers that
    did report is returned with their count."""
    import multiprocessing as mp
    import queue as _queue

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if max_workers:
        n = min(n, max_workers)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(seconds, pl_size, q, mode)) for _ in range(n)]
    for p in procs:
        p.start()
    res, deadline = [], time.monotonic() + seconds * 3 + 90
    try:
        while len(res) < n and time.monotonic() < deadline:
            try:
                res.append(q.get(timeout=1.0))
            except _queue.Empty:
                if not any(p.is_alive() for p in procs) and q.empty():
                    break
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=2)
    return {"value": sum(r[0] for r in res), "unit": "env-steps/s", "updates_per_s": sum(r[1] for r in res), "cores": len(res),
            "sample": f"{len(res)} of {n} independent single-thread processes of the same loop reported, {seconds:.0f} s each, "
                      "throughputs summed"}


# The workload whose throughput is `value` when no --mode is given; the other one is measured in the same run and
# reported under `also_measured`. The same at every N, so that value(N) / value(1) is a scaling figure.
# interfrl = the north star's multi-GPU workload (federated platoons, one RCCL all-reduce of the actor/critic gradients per
# step; BASELINE configs[3] per GPU, configs[1]'s 4096 x 5 shape), run with the f32-class split-operand set learner
# (csrc/fsplit.hip, 1e-4 of max against the float64 oracle like the f32 kernels). nofrl = the reference's default
# `fed_method` (one independent weight / Adam set per agent): HBM-bound at 2.47 MB per agent-update, the r01/r02 headline.
PRIMARY_MODE = "interfrl"
# what the plain 1-GPU default run measures besides the two 4096 x 5 workloads (keys of also_measured; tests/test_abi_cpu.py)
EXTRA_CONFIGS = {
    "config3_interfrl": "BASELINE configs[2]: 4096 platoons x 10 vehicles, 1x MI355X -- interfrl, split engine",
    "config3_nofrl": "BASELINE configs[2]: 4096 platoons x 10 vehicles, 1x MI355X -- nofrl (the reference's default fed_method)",
    "config5": "BASELINE configs[4]: 4096 platoons, actor/critic hidden = 1024, bf16, 1x MI355X -- interfrl, batched engine",
}


def env_overrides():
    """Names of the environment variables that could change what the timed region executes: every AVD_* variable (switches
    of the diagnostic library build, bench-order / pause knobs) and AVDDPG_HIP_LIB (another build of the library). The
    shipped library reads none of them (tests/test_abi_cpu.py); bench.py refuses to run when any is set unless
    --allow-diagnostics, and always prints the list."""
    return sorted(k for k in os.environ if (k.startswith("AVD_") and k != "AVD_BENCH_SPAWN_PROBE") or k == "AVDDPG_HIP_LIB")


def refuse_diagnostics(args):
    ov = env_overrides()
    if ov and not args.allow_diagnostics:
        sys.exit(f"bench.py: refusing to measure with diagnostic environment overrides set: {', '.join(ov)} "
                 "(unset them, or pass --allow-diagnostics to run anyway and have them recorded in the line)")
    return ov


class stdout_to_stderr:
    """Keep rank 0's stdout to the ONE JSON line: gloo prints a connection banner and RCCL a version banner through C stdio while a
    communicator is built (the latter sits in libc's buffer and would surface AFTER the JSON line at exit). Inside the block fd 1 is
    fd 2; on exit libc's buffers are flushed (into stderr) before fd 1 is restored."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one, which has not
    touched the GPU (no torch import, no HIP call) and only waits. Each child is this script with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, i.e. exactly what torch.distributed.run would start. Rank 0's stdout is ours."""
    import socket
    import subprocess

    port, holder = args.master_port, None
    if not port:
        # pick a free port and KEEP it bound (never listening, SO_REUSEADDR) until the ranks are done: the kernel will not hand
        # it to another process's port-0 bind meanwhile, and rank 0's store (which also sets SO_REUSEADDR) can still bind + listen
        holder = socket.socket()
        holder.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        holder.bind(("127.0.0.1", 0))
        port = holder.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:  # one rank failed: the others would wait in a collective forever
                    rc = rc or code
                    for o in pending:
                        o.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if holder is not None:
            holder.close()
    return rc


def build_trainer(args, mode, engine, rank, group, ring=None):
    from avddpg_amd import config, trainer

    P, L = args.platoons, args.pl_size
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=args.buffer_size,
                         fed_method={"interfrl": "interfrl", "intrafrl": "intrafrl"}.get(mode, "normal"),
                         intra_directional_averaging=bool(getattr(args, "directional", False)),
                         weighted_average_enabled=bool(getattr(args, "weighted", False) and mode == "interfrl"), random_seed=1,
                         framework=args.framework)  # random_seed: initial weights, the same on every rank
    if args.hidden:
        conf.actor_layer1_size = conf.actor_layer2_size = conf.critic_layer1_size = conf.critic_layer2_size = args.hidden
    return trainer.VecTrainer(conf, rng="device", group=group if mode == "interfrl" else None, auto_reset=True,
                              seed=1 + rank, pipeline_chunks=(args.intra_chunks if mode == "intrafrl" else args.chunks),
                              fused_update=(mode == "nofrl" and not args.no_fused),
                              shared_engine=engine if mode == "interfrl" else None, replay_ring=ring,
                              overlap_allreduce=True if (args.overlap and mode == "interfrl" and group is not None and engine == "fused3") else None)


def run_workload(args, mode, engine, rank, world, group, vt=None):
    """Time args.steps steps of one workload (trainer `vt`, built here when not given); return the fields of the JSON line."""
    import torch

    P, L = args.platoons, args.pl_size
    if vt is None:
        vt = build_trainer(args, mode, engine, rank, group)
    batched = vt.shared and vt.shared_engine in ("batched", "fused", "fused3")
    fset = vt.shared and vt.shared_engine in ("fused", "fused3")
    split3 = vt.shared and vt.shared_engine == "fused3"
    # synthetic steady state: replay rings full of random-init-platoon-like rows
    ring = vt.replay.ring
    chunk = max(1, (1 << 28) // (ring.shape[1] * ring.shape[2]))
    for a0 in range(0, ring.shape[0], chunk):
        ring[a0:a0 + chunk].normal_(0.0, 1.0)
    vt.replay.buffer_counter = args.buffer_size
    vt.reset_episode()

    # per-stage / per-kernel HIP events, each recorded on the stream its kernels are launched on
    names = ("act+env", "replay", "learn", "update", "learn+update", "allreduce")  # (allreduce: inside "learn"; its own events, on its stream)

    def one_step(record):
        vt.timers = ev if record else None
        vt.step()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(warmup, steps):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize; max over ranks. Returns
        (seconds, per-step summed launch durations of each stage)."""
        nonlocal ev
        ev = {}
        for _ in range(warmup):
            one_step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step(True)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, {n: sum(a.elapsed_time(b) for a, b in ev.get(n, [])) / steps for n in names}

    # steady state first: untimed steps until --prewarm-seconds have passed (clock / power ramp out of idle), THEN the W warm-up steps
    # and the K timed ones
    prewarm_steps, t_pw = 0, time.perf_counter()
    ev = {}
    def more_prewarm():
        go = args.prewarm_seconds > 0 and time.perf_counter() - t_pw < args.prewarm_seconds
        if world > 1:  # every rank must run the SAME number of steps (interfrl steps hold a collective): continue while any rank wants to
            import torch.distributed as dist
            flag = torch.tensor([1 if go else 0], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            go = bool(flag.item())
        return go

    while more_prewarm():
        for _ in range(16):
            one_step(False)
        torch.cuda.synchronize()
        prewarm_steps += 16
    elapsed, stage_ms = timed_run(args.warmup, args.steps)
    # N > 1, split engine: time the OTHER collective form too (same trainer, same state), so that one multi-GPU run shows what the
    # overlap hides: forms.single = one all-reduce of the whole slab between learn and Adam; forms.overlapped = critic block on a side
    # stream under the actor phase + actor block on the main stream. `value` is the form the trainer runs: the single collective
    # unless --overlap (on a one-rank RCCL communicator the overlapped form measured slower).
    forms = None
    if (world > 1 or group is not None) and mode == "interfrl" and split3 and not args.no_collective_ab:
        chosen = "overlapped" if vt.overlap_allreduce else "single"
        forms = {chosen: {"ms_per_step": 1e3 * elapsed / args.steps, "stages_ms": stage_ms}}
        vt.overlap_allreduce = not vt.overlap_allreduce
        alt_steps = max(1, min(args.steps, 500))
        e2, st2 = timed_run(min(args.warmup, 50), alt_steps)
        forms["single" if chosen == "overlapped" else "overlapped"] = {"ms_per_step": 1e3 * e2 / alt_steps, "stages_ms": st2, "steps": alt_steps}
        vt.overlap_allreduce = not vt.overlap_allreduce

    # per step: summed launch durations of each stage (learn/update: all agent slices of the step)
    n_agents = P * vt.M  # models: L per platoon (decentralized), 1 (centralized)
    env_steps_per_s = world * P * args.steps / elapsed
    updates_per_s = world * n_agents * args.steps / elapsed
    fused = stage_ms["learn+update"] > 0
    # algorithmic work per agent-update from the LOGICAL network dims (at the reference widths these are the
    # module constants: 0.751 MFLOP per sample, 2.47 MB per update)
    d = vt.agents.dims
    a_fwd = 2 * (d.S * d.H1 + d.H1 * d.H2 + d.H2 * d.A)
    c_fwd = 2 * (d.S * d.H1 + d.A * d.Ha + (d.H1 + d.Ha) * d.H2 + d.H2 * d.A)
    n_train = (d.S * d.H1 + 3 * d.H1 + d.H1 * d.H2 + 3 * d.H2 + d.H2 * d.A + d.A) + \
              (d.S * d.H1 + 3 * d.H1 + d.A * d.Ha + 3 * d.Ha + (d.H1 + d.Ha) * d.H2 + 3 * d.H2 + d.H2 * d.A + d.A)
    n_stats = 2 * (d.H1 + d.H2) + 2 * (d.H1 + d.Ha + d.H2)
    flop_per_sample = 4 * a_fwd + 6 * c_fwd  # 5 forwards + the backward passes of critic, actor and actor-through-critic
    adam_bytes = 32 * n_train + 12 * n_stats  # r/w of W, W_target, m, v + the soft update of the BN statistics
    if (d.S, d.A, d.H1, d.H2, d.Ha) == (4, 1, 256, 128, 48):
        assert abs(flop_per_sample - LEARN_FLOP_PER_SAMPLE) < 0.01 * LEARN_FLOP_PER_SAMPLE
        assert abs(adam_bytes - ADAM_BYTES_PER_AGENT) < 0.01 * ADAM_BYTES_PER_AGENT
    roofs = []
    # the reference widths run learn_kernel_l (lean.hip: two workgroups per CU) unless AVD_LEARN_KERNEL=fast (learn_kernel_t)
    from avddpg_amd import _hip
    lk = ("learn_kernel_t" if (os.environ.get("AVD_LEARN_KERNEL") == "fast" and _hip.lib().avd_diagnostics_enabled())
          else "learn_kernel_l")
    # learn_kernel_l<fused> updates the small tensors itself (one launch); learn_kernel_t leaves them to a second kernel
    fused_name = lk + "<fused>" + ("" if lk == "learn_kernel_l" else " + adam_polyak_ranges_kernel")
    if args.framework != "decentralized":
        # widths x 1.2 at L = 3 / 5: cen.hip (eight-wave kernel; the update as chunked learn kernels in the caller's stream and their
        # whole-row Adam + Polyak passes on a side stream); any other centralized shape: the general kernel
        lay_ = vt.agents.lay  # (padded widths)
        if (lay_.H1, lay_.H2, lay_.Ha) == (320, 160, 64) and (lay_.S, lay_.A) in ((12, 3), (20, 5)):
            lk, fused_name = "cen::learn_kernel_c", "cen::learn_kernel_c in chunks || adam_polyak_rows_kernel (two streams)"
        else:
            lk, fused_name = "gen::learn_kernel_g", "gen::learn_kernel_g<fused> + adam_polyak_ranges_kernel"
    intra = mode == "intrafrl"
    if intra:
        # learn_kernel_l (gradients to the slab) in platoon chunks || adam_polyak_intra_kernel (the platoon's mean formed where Adam
        # consumes it) on a side stream: the whole region against both roofs, like the fused nofrl kernel
        piped = stage_ms["learn+update"] > 0
        fused_name = (f"learn_kernel_l (gradients out) in {vt.pipeline_chunks} platoon chunks || adam_polyak_intra_kernel (platoon mean + Adam x2 + "
                      "Polyak, one pass over the gradient slab; side stream)") if piped else "learn_kernel_l, then adam_polyak_intra_kernel"
        if not piped:
            stage_ms["learn+update"] = stage_ms["learn"] + stage_ms["update"]
        fused = True
    if fused:
        # one kernel does Trainer.learn AND Adam x2 + Polyak: price it against both roofs, the binding one is the
        # roof it sits closer to
        t = stage_ms["learn+update"] / 1e3
        mf = {"kernel": fused_name, "bound": "mfma", "achieved": flop_per_sample * 64 * n_agents / t / 1e12,
              "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None}
        hb = {"kernel": fused_name, "bound": "hbm", "achieved": adam_bytes * n_agents / t / 1e9,
              "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
        roofs = [mf, hb]
    elif batched:
        # shared-set learner: the GEMM FLOPs of the chain (forward x5, weight-gradient x2, input-gradient x3) over the
        # whole learn stage (GEMMs + the bandwidth-bound row/column kernels between them), against the dense bf16 peak
        lay = vt.agents.lay
        H1, H2, KC, rows = lay.H1, lay.H2, lay.H1 + lay.Ha, 64 * n_agents
        flops = 2.0 * rows * H2 * ((2 * H1 + 3 * KC) + (KC + H1) + (KC + lay.Ha + H1))  # actor fwd x2, critic fwd x3
        learn_s = stage_ms["learn"] / 1e3
        if fset:  # the algorithmic count of SURVEY 8(d) (0.751 MFLOP per sample), like the f32 kernels
            flops = flop_per_sample * 64 * n_agents
        roofs.append({"kernel": ("avd_learn_set_split_f16x3 (fsplit.hip: head x4, actor seed, dw x2, dx x2, dxa persistent kernels + scale, prep, "
                                 "pack, finalize; every operand a 16-bit pair: 2-3 MFMAs per algorithmic product, so the executed "
                                 "matrix work is 2.2x the algorithmic FLOPs priced here: roofline.executed_over_algorithmic)" if split3 else
                                 "avd_learn_set_fused_bf16 (fset.hip: head x6, dw x2, dx x2, dxa persistent kernels + prep, pack, finalize)" if fset else
                                 ("avd_learn_shared_bf16 (wide.hip, hidden >= 512, rank-one backward: fwd_gen x4 -- the critic's two passes of a learn step, "
                                  "(s, a) and (s, mu), are ONE of them --, aux_pack x2, dw_gen x2, w2_post x2, dx_gen x2 + row / table kernels; FLOPs "
                                  "priced = the chain's algorithmic GEMMs, forward x5 incl. the pass that now rides on critic(s, a)'s accumulators)" if lay.H2 >= 512 else "avd_learn_shared_bf16 (gemm_bt256_kernel x10 + row/column kernels)")), "bound": "mfma",
                      "achieved": flops / learn_s / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                      "_t": learn_s})
    else:
        learn_s = stage_ms["learn"] / 1e3
        upd_s = stage_ms["update"] / 1e3
        roofs.append({"kernel": lk, "bound": "mfma", "achieved": flop_per_sample * 64 * n_agents / learn_s / 1e12,
                      "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None, "_t": learn_s})
        if mode == "nofrl":
            roofs.append({"kernel": "adam_polyak_kernel", "bound": "hbm", "achieved": adam_bytes * n_agents / upd_s / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "_t": upd_s})
    # HBM-side bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
    # tools/pmc_workload.py -- the same trainer step at 4096 x L, widths as here --, both counters calibrated on a 1 GiB elementwise
    # kernel of the same pass: tools/pmc_traffic.py; measured factors 2.000 / 1.000). One file per (workload, L, hidden); a line whose
    # shape has no committed pass carries traffic = null.
    def pmc(name):
        for rnd in ("r06", "r05", "r04"):
            path = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
            if os.path.exists(path):
                return json.load(open(path)), f"profiles/{rnd}_{name}"
        return None, None

    def chain_bytes(d, belongs, once_per_learn):
        """bytes per learn of the kernels `belongs` selects, from a profile of several learns (count = launches of `once_per_learn`)"""
        ks = {n: v for n, v in d["kernels"].items() if belongs(n)}
        n_learn = next((v["launches"] for n, v in ks.items() if once_per_learn in n), 0)
        if not n_learn:
            return None
        return sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in ks.values()) / n_learn

    shape_tag = ("" if L == 5 else f"_L{L}") + (f"_h{args.hidden}" if args.hidden else "")
    if P == 4096 and args.framework == "decentralized" and args.chunks == 1:
        cal = lambda d: (d["calibration"]["fetch_factor"], d["calibration"]["write_factor"])
        if fused and mode == "nofrl" and not args.hidden:
            d, src = pmc(f"pmc_traffic_nofrl{shape_tag}.json")
            k = d and next((v for n, v in d["kernels"].items() if "learn_kernel_l" in n), None)
            if k:
                for r in roofs:
                    r["traffic"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
                    r["traffic_note"] = ("HBM-side bytes per launch of learn_kernel_l<fused>, PMC FETCH_SIZE x %.3f + WRITE_SIZE x %.3f (factors "
                                         "calibrated on a 1 GiB elementwise kernel in the same pass), %s" % (*cal(d), src))
        if mode == "interfrl" and split3:
            d, src = pmc(f"pmc_traffic_interfrl{shape_tag}.json")
            t_ = d and chain_bytes(d, lambda n: "fsplit" in n or "finalize" in n, "dxa_kernel")
            if t_:
                roofs[0]["traffic"] = t_
                roofs[0]["traffic_note"] = ("HBM-side bytes per learn over all launches of the chain, PMC FETCH_SIZE x %.3f + WRITE_SIZE x %.3f (calibrated in "
                                            "the same pass), %s (matrix-core bound: reported, not the binding roof)" % (*cal(d), src))
        if intra and not args.hidden:
            d, src = pmc(f"pmc_traffic_intrafrl{shape_tag}.json")
            if d:
                n_steps = next((v["launches"] for n, v in d["kernels"].items() if "step_fused_kernel" in n), 0)
                ks = [v for n, v in d["kernels"].items() if "learn_kernel_l" in n or "adam_polyak_intra" in n or "polyak_intra" in n]
                if n_steps and ks:
                    per_step = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in ks) / n_steps
                    for r in roofs:
                        r["traffic"] = per_step
                        r["traffic_note"] = ("HBM-side bytes per step over the learn chunks and the mean + Adam + Polyak passes, PMC FETCH_SIZE x %.3f + "
                                             "WRITE_SIZE x %.3f (calibrated in the same pass), %s" % (*cal(d), src))
        if mode == "interfrl" and batched and not fset and args.hidden:
            d, src = pmc(f"pmc_traffic_interfrl{shape_tag}.json")
            t_ = d and chain_bytes(d, lambda n: "fw::" in n or "wide::" in n, "losses_kernel")
            if t_:
                roofs[0]["traffic"] = t_
                roofs[0]["traffic_note"] = ("HBM-side bytes per learn over all launches of the wide.hip chain, PMC FETCH_SIZE x %.3f + WRITE_SIZE x %.3f "
                                            "(calibrated in the same pass), %s" % (*cal(d), src))
    # the centralized pipeline as the library will cut it on THIS device (avd_learn_update_plan: chunk = one learn workgroup per CU,
    # at most 32 chunks), not literals of the profiled box
    plan = None
    if args.framework == "centralized" and fused and lk == "cen::learn_kernel_c":
        import ctypes
        ch, nch, grp_ = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _hip.call("avd_learn_update_plan", ctypes.byref(vt.agents.lay), n_agents, ctypes.byref(ch), ctypes.byref(nch), ctypes.byref(grp_))
        plan = {"chunk_agents": ch.value, "chunks": nch.value, "update_workgroups": grp_.value}
        fused_name = f"cen::learn_kernel_c x{nch.value} chunks of {ch.value} || adam_polyak_rows_kernel ({grp_.value} workgroups, side stream)"
        for r in roofs:
            r["kernel"] = fused_name
    if P == 4096 and L == 5 and plan and pmc("pmc_traffic_centralized.json")[0]:
        d, src = pmc("pmc_traffic_centralized.json")
        ks = [v for n, v in d["kernels"].items() if "learn_kernel_c" in n or "adam_polyak_rows" in n]
        if len(ks) == 2 and plan["chunk_agents"] == 256:  # (the profiled launches are chunks of 256 models: valid for that plan only)
            per_step = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in ks) * plan["chunks"]
            alg = adam_bytes * n_agents
            for r in roofs:
                r["traffic"] = per_step
                r["traffic_note"] = ("HBM-side bytes per step = %d x (learn chunk + update pass), PMC FETCH_SIZE x %.3f + WRITE_SIZE x %.3f (calibrated on a "
                                     "1 GiB elementwise kernel in the same pass; kernels serialised by the counter pass), %s: "
                                     "%.1f x the %.1f GB priced here -- the gradients' round trip through HBM and re-reads of the online W2 matrices that miss L2"
                                     % (plan["chunks"], d["calibration"]["fetch_factor"], d["calibration"]["write_factor"], src, per_step / alg, alg / 1e9))
    for r in roofs:
        r["frac"] = r["achieved"] / r["peak"]
    if fused:
        dominant = max(roofs, key=lambda r: r["frac"])
    else:
        dominant = max(roofs, key=lambda r: r.pop("_t"))
        for r in roofs:
            r.pop("_t", None)
    roof_learn, roof_upd = roofs[0], (roofs[1] if len(roofs) > 1 else None)
    split_extra = {}
    if split3:
        # executed matrix work: the MFMA instructions the chain issues (from the kernels' loop structure, C ABI) x 32 768 FLOP
        import ctypes
        from avddpg_amd import _hip as hip_
        cnt = ctypes.c_ulonglong(0)
        hip_.call("avd_learn_set_split_mfma_count", ctypes.byref(vt.agents.lay), n_agents, vt.M, ctypes.byref(cnt))
        executed = cnt.value * 32768.0
        ratio = executed / (flop_per_sample * 64 * n_agents)
        split_extra = {"executed_mfma_32x32x16_per_learn": cnt.value, "executed_over_algorithmic": ratio,
                       "executed_frac": ratio * dominant["frac"],
                       "power": "the chain is paced by the clock governor, not by idle cycles: rocm-smi during 2500 back-to-back learns reads "
                                "2.10-2.12 GHz at 1.34-1.36 kW on one box (below the 1400 W cap; an r04 box sat at 1398-1399 W), and a build whose dx "
                                "idles less is clocked 4.4 % lower (profiles/r05_power_clock_ab_product_vs_two_tile_dx.txt); dense fp16 MFMA holds "
                                "2.3 PFLOP/s with one operand static, 1.30-1.35 PFLOP/s with both operands fresh from LDS on every MFMA "
                                "(profiles/r03_power_*.txt, r04_mfma_shape_probe.txt)"}
    rccl = mode == "interfrl" and (world > 1 or group is not None)

    out = {
        "value": env_steps_per_s,
        "unit": "env-steps/s",
        "updates_per_s": updates_per_s,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "dtype": ("f32-class (fp16 hi+lo operand pairs on the 16-bit matrix cores, f32 accumulate)" if split3 else
                  "bf16 (GEMM operands; f32 accumulation, parameters and optimiser)" if batched else "f32"),
        "config": {"workload": f"{P} platoons x {L} vehicles per GPU, DDPG "
                               + ("centralized " if args.framework == "centralized" else "")
                               + ("nofrl: one actor/critic/target/Adam/replay set per (platoon, vehicle)"
                                  if mode == "nofrl" else
                                  ("intrafrl+gradients: one set per (platoon, vehicle), every agent stepping with the mean gradient of its platoon's "
                                   "vehicles (workers/trainer.py:189-190, 417-431)" + (", directional (lead vehicle skipped)" if args.directional else ""))
                                  if mode == "intrafrl" else
                                  "interfrl+gradients: one weight set per vehicle index, RCCL all-reduce of grads"
                                  + (f", engine={vt.shared_engine}" + (f", hidden={args.hidden}" if args.hidden else ""))
                                  + (", WEIGHTED mean (|1 / mean of the last 10 episodic rewards| per agent, weights on the device)"
                                     if getattr(vt, "_dev_weighted", False) else ""))
                               + f", B=64, replay capacity {args.buffer_size} (full), 1 update per env step",
                   "platoons_per_gpu": P, "pl_size": L, "agents_per_gpu": n_agents, "mode": mode,
                   "parallelism": f"platoon shards x{world}" + (f" + one {args.backend} all-reduce(sum) of the [M, theta] gradient slab per step"
                                                                if rccl else " (no data-path collective)")},
        "dtype_note": ("every matrix-product operand an fp16 pair hi+lo (hi = rn16(x), lo = rn16(x - hi): worst-case residual 2^-22 |x|, "
                       "measured worst 2^-23; r03's bf16 pairs -- 2^-17 worst -- are gone), scaled by exact powers of two into fp16's range, "
                       "times exact +-1/0 relu masks where the algebra has them; A_hi B_hi + A_lo B_hi + A_hi B_lo, f32 accumulation, parameters, "
                       "gradients and optimiser. Enforced by the GPU tests at 2e-5 of each gradient tensor's max against the float64 oracle "
                       "(4e-6 at <= 4480 rows; the exact-f32 kernels: 1e-4), incl. two whole sets at 4096 x 5, and by a 2036-update reward-curve "
                       "test against the float32 noise floor (tests/test_gpu_fsplit.py, tests/test_gpu_configs_full.py); that it TRAINS: 50 k "
                       "updates per set at this shape take the evaluator score from -248 to -4 (DESIGN.md 5.1, profiles/r05_training_curves/)" if split3 else None),
        "roofline": {k: dominant[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | {"kernel": dominant["kernel"]}
                    | (split_extra if split3 else {}),
        "stages_ms": stage_ms,
        "prewarm": {"steps": prewarm_steps, "seconds": args.prewarm_seconds,
                    "note": "untimed steps of the same workload before the --warmup steps (steady clocks / power state); not in `steps`, `warmup` or the timed region"},
        "pipeline": (fused_name if intra else
                     (f"fused learn+Adam+Polyak kernel (avd_learn_update_act_f32 -> {lk}); the NEXT step's actor forward is "
                      "evaluated in that kernel's epilogue on the weights it has just written (same values as the separate "
                      "actor launch, which now runs only after an episode reset): its time is inside learn+update, not act+env")
                     if (fused and getattr(vt, "_act_ready", False)) else
                     (f"avd_learn_update_f32 -> cen.hip: the agents in {plan['chunks'] if plan else '?'} chunks of {plan['chunk_agents'] if plan else '?'}; chunk c's learn kernel (cen::learn_kernel_c, in the "
                      "caller's stream) is followed by its whole-row Adam + Polyak pass (adam_polyak_rows_kernel, a side stream) under chunk "
                      "c + 1's learn kernel") if (fused and lk == "cen::learn_kernel_c") else
                     f"fused learn+Adam+Polyak kernel (avd_learn_update_f32 -> {lk})" if fused else
                     (f"learn || Adam+Polyak over {args.chunks} agent slices on 2 HIP streams; stage times are "
                      "summed kernel durations and overlap") if (mode == "nofrl" and args.chunks > 1) else
                     ("split-operand fused shared-set learner (avd_learn_set_split_f16x3) + Adam/Polyak on the sets" if split3 else
                      "fused shared-set learner (avd_learn_set_fused_bf16: second-layer weights resident in registers, first layers on "
                      "the matrix cores, per-workgroup gradient partials) + Adam/Polyak on the sets" if fset else
                      "batched shared-set learner (avd_learn_shared_bf16) + Adam/Polyak on the sets" if batched else "serial")),
        "kernels": [r for r in (roof_learn, roof_upd) if r],
    }
    if plan:
        out["update_plan"] = plan
    if rccl:
        out["rccl_ranks"] = world if args.backend == "nccl" else 0
        out["collective_backend"] = args.backend
        out["collective"] = {"per_step_ms": stage_ms["allreduce"], "overlapped": bool(getattr(vt, "overlap_allreduce", False)),
                             "overlap_is_real": bool(getattr(vt, "overlap_allreduce", False)) and args.backend == "nccl",
                             "forms": forms,
                             "hidden_ms": ((forms["single"]["ms_per_step"] - forms["overlapped"]["ms_per_step"]) if forms else None),
                             "bytes_per_step": 4 * vt.M * vt.agents.lay.theta_size,
                             "note": ("two all-reduce(sum) per step: the critic block [M, 41 412] on a side stream under the actor phase of the learn "
                                      "call, the actor block [M, 35 076] on the main stream; per_step_ms = their summed durations on their own streams "
                                      "(inside stages_ms.learn only as far as they are not hidden)") if getattr(vt, "overlap_allreduce", False) else
                                     "one all-reduce(sum) of the [M, theta] slab per step, on the main stream between learn and Adam (inside stages_ms.learn)"}
    del vt
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if not os.environ.get("AVD_BENCH_SPAWN_PROBE"):
        overrides = refuse_diagnostics(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # this process never touches the GPU
    if os.environ.get("AVD_BENCH_SPAWN_PROBE"):  # launcher plumbing check (tests/test_dist_cpu.py): no GPU, no torch
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        sys.exit(0 if os.environ.get("AVD_BENCH_SPAWN_PROBE") != "fail1" or os.environ.get("RANK") != "1" else 3)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1:  # (a launcher's WORLD_SIZE with --gpus left at its default is taken as N; a contradiction is not)
            sys.exit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
        args.gpus = world
    from avddpg_amd import _hip
    diag_build = bool(_hip.lib().avd_diagnostics_enabled())
    if diag_build and not args.allow_diagnostics:
        sys.exit(f"bench.py: {_hip.LIB_PATH} is a diagnostic build (-DAVD_DIAG); pass --allow-diagnostics to measure it anyway")
    dev = 0 if args.single_device else local_rank
    # fail fast, non-zero and with a reason (the parent of self-spawned ranks then stops the others): fewer GPUs than ranks
    # (device_count() does not initialise the GPU), a rendezvous / first collective that does not complete in --init-timeout
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (not args.single_device and n_dev < world):
        sys.exit(f"bench.py: --gpus {world} needs {world} visible GPUs, torch.cuda.device_count() = {n_dev} "
                 "(one rank per GPU; --single-device puts every rank on GPU 0 for plumbing tests only)")
    torch.cuda.set_device(dev)
    group = None
    if world == 1 and args.one_rank_rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port or 29517))
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev))
            hello = torch.ones(1, device="cuda")
            dist.all_reduce(hello)
            torch.cuda.synchronize()
        group = dist.group.WORLD
    if world > 1:
        import datetime
        import threading

        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

        def give_up():
            sys.stderr.write(f"bench.py: rank {rank}: process-group init + first collective over {args.backend} did not complete "
                             f"within --init-timeout {args.init_timeout:.0f} s (MASTER_ADDR={os.environ.get('MASTER_ADDR')} "
                             f"MASTER_PORT={os.environ.get('MASTER_PORT')}); giving up\n")
            sys.stderr.flush()
            os._exit(4)

        watchdog = threading.Timer(args.init_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        timeout = datetime.timedelta(seconds=max(30.0, args.init_timeout))
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=timeout)
                hello = torch.ones(1, device="cuda")
                dist.all_reduce(hello)  # RCCL builds its communicator on the first collective: do it here, under the watchdog
                torch.cuda.synchronize()
                if int(hello.item()) != world:
                    sys.exit(f"bench.py: first all-reduce over RCCL returned {hello.item()} on rank {rank}, expected {world}")
            else:
                dist.init_process_group(args.backend, timeout=timeout)
                dist.barrier()
        watchdog.cancel()
        group = dist.group.WORLD

    if args.mode:
        modes = [args.mode]
    elif args.framework != "decentralized":
        # the centralized framework has no federated method (src/config.py:26-27: fed_enabled needs the decentralized one): one model
        # per platoon, the fused learn + Adam + Polyak kernel
        modes = ["nofrl"]
    else:
        modes = [PRIMARY_MODE] + ([] if args.no_secondary or args.hidden or args.framework != "decentralized" else
                                  ["interfrl" if PRIMARY_MODE == "nofrl" else "nofrl"])
    # interfrl without --engine: the f32-class fused set learner where it exists (reference widths, decentralized)
    engine = args.engine or ("fused3" if (not args.hidden and args.framework == "decentralized") else None)
    if os.environ.get("AVD_BENCH_ORDER") == "rev":  # diagnostics: measure the secondary workload first
        modes = modes[::-1]
    # Two workloads: BOTH trainers are built before anything is timed -- the one with per-agent slabs (nofrl: 25 GB of
    # weights / moments + the 82 GB replay ring) first, on a pristine heap, and the other one sharing its ring. Building the
    # second trainer after freeing the first one's 100+ GB left it ~5 % slow (page placement of the re-allocated slabs).
    trainers = {}
    if len(modes) == 2:
        first = "nofrl" if "nofrl" in modes else modes[0]
        trainers[first] = build_trainer(args, first, engine, rank, group)
        other = modes[1] if modes[0] == first else modes[0]
        trainers[other] = build_trainer(args, other, engine, rank, group, ring=trainers[first].replay.ring)
    results = []
    for i, m in enumerate(modes):
        if i:
            time.sleep(float(os.environ.get("AVD_BENCH_PAUSE", "1.0")))  # idle gap, outside both timed regions
        results.append(run_workload(args, m, engine, rank, world, group, vt=trainers.pop(m, None)))
    if os.environ.get("AVD_BENCH_ORDER") == "rev":
        results = results[::-1]

    # The other single-GPU BASELINE configs under the same clock: configs[2] = 4096 x 10 (interfrl with the split
    # engine + the reference's default nofrl, replay capacity 100000) and configs[4] = hidden 1024 (bf16 batched engine), each with its
    # own trainer, prewarm, warm-up and timed region; only in the plain default run on one GPU (no --mode / --hidden / --engine /
    # shape flags), after the two 4096 x 5 workloads have been freed.
    extras = {}
    default_run = (world == 1 and group is None and not args.mode and not args.hidden and not args.engine and not args.no_secondary
                   and not args.weighted
                   and args.framework == "decentralized" and args.platoons == 4096 and args.pl_size == 5 and args.chunks == 1
                   and not args.no_fused and not args.no_extra_configs)
    if default_run:
        import gc
        del trainers
        gc.collect()
        torch.cuda.empty_cache()
        sub = lambda **kw: argparse.Namespace(**{**vars(args), "steps": min(args.steps, 300), "warmup": min(args.warmup, 50), **kw})
        a3 = sub(pl_size=10)
        t_nofrl = build_trainer(a3, "nofrl", None, rank, None)  # (per-agent slabs + the 164 GB ring first, on a clean heap)
        t_inter = build_trainer(a3, "interfrl", "fused3", rank, None, ring=t_nofrl.replay.ring)
        extras["config3_interfrl"] = run_workload(a3, "interfrl", "fused3", rank, world, None, vt=t_inter)
        del t_inter
        extras["config3_nofrl"] = run_workload(a3, "nofrl", None, rank, world, None, vt=t_nofrl)
        del t_nofrl
        gc.collect()
        torch.cuda.empty_cache()
        a5 = sub(hidden=1024)
        extras["config5"] = run_workload(a5, "interfrl", None, rank, world, None)
        assert set(extras) == set(EXTRA_CONFIGS)
        for k, name in EXTRA_CONFIGS.items():
            extras[k]["config"]["baseline_config"] = name
            extras[k]["steps"], extras[k]["warmup"] = a3.steps, a3.warmup

    if rank == 0:
        first = results[0]
        out = {"metric": "env-steps/sec + DDPG updates/sec, 4096x5-vehicle platoons", "value": first["value"],
               "unit": first["unit"], "updates_per_s": first["updates_per_s"], "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": first["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": first["dtype"], "data": "synthetic",
               "primary_mode": first["config"]["mode"],
               # self-certification: the environment overrides in effect (none unless --allow-diagnostics) and the library build
               "env_overrides": overrides, "diagnostic_library": diag_build}
        out.update({k: v for k, v in first.items() if k not in out})
        if len(results) > 1:
            out["also_measured"] = {r["config"]["mode"]: {k: r[k] for k in ("value", "unit", "updates_per_s", "ms_per_step", "dtype",
                                                                          "config", "roofline", "stages_ms", "pipeline")
                                                          + (("rccl_ranks", "collective_backend", "collective") if "rccl_ranks" in r else ())}
                                    for r in results[1:]}
        if extras:
            out.setdefault("also_measured", {}).update(
                {k: {f: r[f] for f in ("value", "unit", "updates_per_s", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline",
                                       "stages_ms", "prewarm", "pipeline")} for k, r in extras.items()})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.pl_size, first["config"]["mode"])
            if args.cpu_all_cores:  # the whole host beside the reference's own 1-thread setting (opt-in side figure)
                try:
                    out["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(min(args.cpu_seconds, 8.0), args.pl_size, mode=first["config"]["mode"])
                except Exception as e:  # never let the side figure break the bench line
                    out["cpu_baseline"]["all_cores"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or group is not None:
        import torch.distributed as dist
        with stdout_to_stderr():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
