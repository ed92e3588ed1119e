"""Multi-GPU plumbing: platoon sharding and the one real exchange step of the hot path.

The unit of independence is a platoon (reference src/environment.py:209-241 touches only its own
followers), so platoons are sharded contiguously over ranks, one process per GPU:
  * nofrl / intrafrl  -- replicas only, no data-path collective;
  * interfrl          -- per federated step ONE all-reduce(sum) of the flat per-vehicle-index gradient slab
                         [M, theta_size] float32 (+ [M] weight sums when weighted), i.e. the cross-platoon
                         mean of reference src/server/federated.py:47-63 / :99-118 with the list-of-platoons
                         axis spread over ranks;
  * any-terminal rule -- the reference ends the episode of ALL platoons when any platoon is terminal (workers/trainer.py:268-269):
                         in the throughput mode the rank's flag rides in the gradient all-reduce (exchange_set_slab), in parity mode
                         (host episode loop) and on steps without an exchange it is a 1-int all-reduce(max).
These helpers contain no device code: they work on CUDA tensors over RCCL ("nccl" backend) and on CPU
tensors over gloo (tests/test_dist_cpu.py)."""
import torch


def shard_platoons(total_platoons, world_size, rank):
    """Contiguous platoon range [lo, hi) owned by `rank`; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(total_platoons, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def total_platoons(n_local, group, device=None):
    """Sum over ranks of the platoons each holds (ranks may hold different counts). A constant of the job: reduce it
    ONCE (trainer construction) and pass it to exchange_fed_sums as `total`; no collective, no host sync per step."""
    import torch.distributed as dist

    if group is None:
        return float(n_local)
    cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return float(cnt.item())


def exchange_fed_sums(out_sum, wsum, n_local, group, total=None):
    """All-reduce the local partial sums of the interfrl average in place and return the divisor of the
    unweighted mean (total number of platoons). out_sum [M, n]; wsum [M] or None; n_local = platoons here.
    ONE collective per call: with weights, the [M] weight sums ride in the same buffer as the slab. `total` = the cached
    result of total_platoons(); without it the count is reduced here (one more collective and a host sync)."""
    import torch.distributed as dist

    if group is None:
        return float(n_local)
    if wsum is None:
        dist.all_reduce(out_sum, op=dist.ReduceOp.SUM, group=group)
    else:
        packed = torch.cat([out_sum.reshape(-1), wsum.reshape(-1).to(out_sum.dtype)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        out_sum.copy_(packed[:out_sum.numel()].view_as(out_sum))
        wsum.copy_(packed[out_sum.numel():].view_as(wsum))
    return total_platoons(n_local, group, out_sum.device) if total is None else float(total)


def set_exchange_buffer(M, theta_size, device):
    """The flat float32 buffer the shared-set learners' gradients are exchanged in: [M * theta] slab | 1 any-terminal flag |
    [M] weight sums -- one allocation for the life of the trainer (the slab view is what the learn call writes), so a federated
    step neither allocates nor copies the slab. Returns (buffer, slab view [M, theta])."""
    buf = torch.zeros(M * theta_size + 1 + M, dtype=torch.float32, device=device)
    return buf, buf[:M * theta_size].view(M, theta_size)


def exchange_set_slab(buf, M, theta_size, wsum, n_local, total, group, flag=None, equal_shards=False):
    """The ONE collective of an interfrl step with shared weight sets (workers/trainer.py:400-431 over platoon shards): all-reduce(sum)
    of `buf` from set_exchange_buffer(), whose slab holds this rank's MEAN gradient per set (what the set learners write).
      * unweighted, every rank holding the same number of platoons (`equal_shards`): the means are summed and scaled by
        n_local / total = 1 / ranks afterwards -- one elementwise launch; otherwise local mean x n_local -> sum -> / total;
      * weighted (`wsum` [M] = this rank's sum of weights per set, src/server/federated.py:99-118): local weighted mean x wsum,
        the [M] sums in the same buffer, divided by the reduced sums afterwards (`wsum` receives them);
      * `flag` (int32[1], this rank's any-terminal flag of the step): rides in the same buffer; afterwards it is non-zero on EVERY
        rank if any rank's was -- the reference ends the episode of ALL platoons when any platoon is terminal
        (workers/trainer.py:268-269), and the conditional reset of the throughput mode reads this flag on the device. No extra
        collective, no host synchronisation.
    Works on CUDA tensors over RCCL and on CPU tensors over gloo."""
    import torch.distributed as dist

    n = M * theta_size
    g = buf[:n].view(M, theta_size)
    if flag is not None:
        buf[n:n + 1].copy_(flag)
    else:
        buf[n:n + 1].zero_()
    if wsum is None:
        if not equal_shards:
            g.mul_(float(n_local))
        dist.all_reduce(buf[:n + 1], op=dist.ReduceOp.SUM, group=group)
        g.mul_(float(n_local) / float(total) if equal_shards else 1.0 / float(total))
    else:
        g.mul_(wsum.view(M, 1))
        buf[n + 1:n + 1 + M].copy_(wsum)
        dist.all_reduce(buf[:n + 1 + M], op=dist.ReduceOp.SUM, group=group)
        wsum.copy_(buf[n + 1:n + 1 + M])
        g.div_(wsum.view(M, 1))
    if flag is not None:
        flag.copy_(buf[n:n + 1] > 0)


def exchange_two_phase(set_grads, actor_size, scale, wsum, total, group, bufs, actor_phase, timers=None, flag=None):
    """The interfrl exchange of a learn call that runs in two phases (avd_learn_set_split_critic / _actor): the CRITIC block
    set_grads[:, actor_size:] is final when this is called and is all-reduced (as the local sum: x `scale`) while
    ``actor_phase()`` -- which must leave the actor block set_grads[:, :actor_size] final -- computes; the actor block (with the
    [M] weight sums of a weighted mean riding in the same buffer) follows; both are divided by `total` / the reduced weight sums
    and written back. Same elementwise sums as exchange_fed_sums on the whole slab (workers/trainer.py:400-431 averages the
    critic and the actor gradient lists independently; src/server/federated.py:47-63).

    Collective ORDER: critic block first, actor block second, both issued from this one host thread with async_op=True -- the
    same sequence on every rank whatever the ranks' timing, which is what a NCCL / RCCL communicator requires (collectives of
    one communicator must be issued in the same order everywhere); waited for in that order too. On CUDA tensors the critic
    block's scaling and collective are queued on bufs["stream"] (a side stream) behind bufs["ready"], so they run under the
    actor phase on the caller's stream; on CPU tensors (gloo tests) the same calls run without streams.
    bufs: dict(crit=[M, T - A] buffer, act=flat [M * A + M + 1] buffer[, stream, ready]). `flag` (int32[1]): this rank's any-terminal
    flag rides behind the actor block (and the weight sums) and comes back non-zero everywhere if any rank's was (exchange_set_slab)."""
    import torch.distributed as dist

    M, A = set_grads.shape[0], actor_size
    cuda = set_grads.is_cuda
    crit, actbuf = bufs["crit"], bufs["act"]
    ev = None
    if cuda:
        main, side = torch.cuda.current_stream(), bufs["stream"]
        bufs["ready"].record(main)
        if timers is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(side):
            side.wait_event(bufs["ready"])
            torch.mul(set_grads[:, A:], scale, out=crit)  # local (weighted) mean -> local (weighted) sum
            if ev:
                ev[0].record()
            # async_op: the host returns at once on every backend, so the actor phase is launched while the collective is in
            # flight (a blocking call would hold the host until a host-staged backend has finished: nothing would overlap)
            work_c = dist.all_reduce(crit, op=dist.ReduceOp.SUM, group=group, async_op=True)
    else:
        torch.mul(set_grads[:, A:], scale, out=crit)
        work_c = dist.all_reduce(crit, op=dist.ReduceOp.SUM, group=group, async_op=True)
    actor_phase()
    act = actbuf[:M * A].view(M, A)
    torch.mul(set_grads[:, :A], scale, out=act)
    if wsum is not None:
        actbuf[M * A:M * A + M].copy_(wsum)
    n = M * A + (M if wsum is not None else 0)
    if flag is not None:
        actbuf[n:n + 1].copy_(flag)
        n += 1
    if ev:
        ev[2].record()
    work_a = dist.all_reduce(actbuf[:n], op=dist.ReduceOp.SUM, group=group, async_op=True)
    if cuda:
        with torch.cuda.stream(side):
            work_c.wait()
            if ev:
                ev[1].record()
        work_a.wait()
        if ev:
            ev[3].record()
            timers.setdefault("allreduce", []).extend([(ev[0], ev[1]), (ev[2], ev[3])])
        main.wait_stream(side)
    else:
        work_c.wait()
        work_a.wait()
    div = float(total) if wsum is None else actbuf[M * A:M * A + M].view(M, 1)
    set_grads[:, A:].copy_(crit.div_(div))
    set_grads[:, :A].copy_(act.div_(div))
    if flag is not None:
        flag.copy_(actbuf[n - 1:n] > 0)


def broadcast_agents(agents, group, src=0):
    """Every rank starts from rank `src`'s initial weights: the reference starts every agent from agent (0,0)'s weights
    and the targets from their online nets (workers/trainer.py:121-131), and the shared-set / interfrl arithmetic
    relies on all ranks holding identical sets. Called right after construction, when every set of `agents` is still
    a copy of set 0 and the Adam state is zero: one set travels, each rank replicates it.
    `agents`: any object with theta, stats, theta_t, stats_t tensors of shape [n_sets, size]."""
    import torch.distributed as dist

    if group is None:
        return
    root = dist.get_global_rank(group, src)
    for name in ("theta", "stats", "theta_t", "stats_t"):
        x = getattr(agents, name)
        first = x[0:1].contiguous()
        dist.broadcast(first, src=root, group=group)
        x.copy_(first.expand_as(x))


def any_terminal(flag, group):
    """Episode ends for ALL platoons on every rank when any platoon anywhere is terminal."""
    import torch.distributed as dist

    if group is None:
        return bool(flag.item())
    f = flag.clone()
    dist.all_reduce(f, op=dist.ReduceOp.MAX, group=group)
    return bool(f.item())
