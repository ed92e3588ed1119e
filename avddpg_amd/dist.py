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


def exchange_fed_sums(out_sum, wsum, n_local, group):
    """All-reduce the local partial sums of the interfrl average in place and return the divisor of the
    unweighted mean (total number of platoons). out_sum [M, n]; wsum [M] or None; n_local = platoons here."""
    import torch.distributed as dist

    if group is None:
        return float(n_local)
    dist.all_reduce(out_sum, op=dist.ReduceOp.SUM, group=group)
    if wsum is not None:
        dist.all_reduce(wsum, op=dist.ReduceOp.SUM, group=group)
    cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=out_sum.device)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)  # ranks may hold different platoon counts
    return float(cnt.item())


def any_terminal(flag, group):
    """Episode ends for ALL platoons on every rank when any platoon anywhere is terminal."""
    import torch.distributed as dist

    if group is None:
        return bool(flag.item())
    f = flag.clone()
    dist.all_reduce(f, op=dist.ReduceOp.MAX, group=group)
    return bool(f.item())
