"""Multi-GPU plumbing: platoon sharding and the one real exchange step of the hot path.

The unit of independence is a platoon (reference src/environment.py:209-241 touches only its own
followers), so platoons are sharded contiguously over ranks, one process per GPU:
  * nofrl / intrafrl  -- replicas only, no data-path collective;
  * interfrl          -- per federated step ONE all-reduce(sum) of the flat per-vehicle-index gradient slab
                         [M, theta_size] float32 (+ [M] weight sums when weighted), i.e. the cross-platoon
                         mean of reference src/server/federated.py:47-63 / :99-118 with the list-of-platoons
                         axis spread over ranks;
  * parity mode       -- a 1-int all-reduce(max) of the any-terminal flag (workers/trainer.py:268-269).
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
