"""User sharding for multi-GPU serving.

The reference's only parallelism on this path is a split of users over worker threads, contiguous
ranges (tdm/src/main/scala/com/mass/tdm/evaluation/Evaluator.scala:28-37).  Here a worker is one
process per GPU; table and weights are replicated, users are the only sharded axis, and there is no
data-path collective: torch.distributed (RCCL on GPUs, gloo in CPU tests) carries only the barrier,
the max-over-ranks clock and an optional gather of results.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks take the extra."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend=None):
    """Returns (dist module or None, rank, world, local_rank) from the torchrun environment."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return None, rank, world, local
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = os.environ.get("DM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    return dist, rank, world, local


def max_over_ranks(value, dist):
    """Wall time of the slowest rank (the job's time)."""
    if dist is None:
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local_ids, dist):
    """Concatenate every rank's [U_r, k] id block in rank order (rank 0 receives the full array)."""
    if dist is None:
        return np.asarray(local_ids)
    import torch
    objs = [None] * dist.get_world_size()
    dist.all_gather_object(objs, np.asarray(local_ids))
    return np.concatenate(objs, axis=0)


def sharded_rows(compute, n_items, dist):
    """Item-sharded evaluation of a per-item row function (JTM child weights): rank r computes
    compute(lo, hi) -> [hi - lo, ...] for its contiguous item range and every rank receives the concatenation in item
    order.  Each item's row is produced by exactly one rank with that rank's own sequential sums, so the result is
    bit-identical to the single-rank run whatever the world size."""
    if dist is None:
        return np.asarray(compute(0, int(n_items)))
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(n_items, rank, world)
    mine = np.ascontiguousarray(compute(lo, hi))
    objs = [None] * world
    dist.all_gather_object(objs, mine)
    return np.concatenate(objs, axis=0)
