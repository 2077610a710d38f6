"""User sharding for multi-GPU serving and item sharding for JTM.

The reference's only parallelism on this path is a split of users over worker threads, contiguous
ranges (tdm/src/main/scala/com/mass/tdm/evaluation/Evaluator.scala:28-37).  Here a worker is one
process per GPU; table and weights are replicated, users are the only sharded axis, and there is no
data-path collective.  The helpers below take a `comm`: dismember_amd.comm.Comm (the library's RCCL /
host transport) or any object with `.rank`, `.world`, `.all_gather_array(a)`, `.allreduce(v, op)`
(the CPU tests pass a gloo-backed adapter); comm=None is the single-worker case.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks take the extra."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, comm):
    """Wall time of the slowest rank (the job's time)."""
    if comm is None:
        return float(value)
    return float(comm.allreduce(float(value), op="max"))


def gather_results(local_ids, comm):
    """Concatenate every rank's [U_r, k] id block in rank order."""
    if comm is None:
        return np.asarray(local_ids)
    return comm.all_gather_array(np.asarray(local_ids))


def sharded_rows(compute, n_items, comm):
    """Item-sharded evaluation of a per-item row function (JTM child weights): rank r computes
    compute(lo, hi) -> [hi - lo, ...] for its contiguous item range and every rank receives the concatenation in item
    order.  Each item's row is produced by exactly one rank with that rank's own sequential sums, so the result is
    bit-identical to the single-rank run whatever the world size."""
    if comm is None:
        return np.asarray(compute(0, int(n_items)))
    lo, hi = shard_range(n_items, comm.rank, comm.world)
    return comm.all_gather_array(np.ascontiguousarray(compute(lo, hi)))
