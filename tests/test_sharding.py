"""N>1 path on CPU: two gloo processes shard users, agree on the slowest-rank clock and
reassemble results in user order (what bench.py --gpus N does with RCCL on GPUs)."""
import os
import socket

import numpy as np
import pytest

from dismember_amd.sharding import shard_range


def test_shard_range_partitions_contiguously():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dismember_amd import sharding
    dist, r, w, _ = sharding.init_distributed("gloo")
    n_users = 101
    lo, hi = sharding.shard_range(n_users, r, w)
    # stand-in for the per-rank beam search: "ids" derived from the global user index
    local = np.stack([np.arange(lo, hi) * 10 + k for k in range(3)], axis=1).astype(np.int32)
    dist.barrier()
    slowest = sharding.max_over_ranks(1.0 + r, dist)
    allids = sharding.gather_results(local, dist)
    q.put((r, slowest, allids.shape, bool((allids[:, 0] == np.arange(n_users) * 10).all())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r, slowest, shape, ordered in out:
        assert slowest == 2.0 and shape == (101, 3) and ordered
