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


# ---- gradient exchange protocol of the data-parallel trainer (3 gloo ranks, torch-backed fake engines) ----
class _FakePort:
    """Stands in for EngineGradPort: a dense block and a row-sparse embedding gradient held in torch CPU tensors."""

    def __init__(self, torch, rank, n_rows=50, E=4):
        g = torch.Generator().manual_seed(100 + rank)
        self.torch = torch
        self.dense_block = torch.randn(37, generator=g)
        self.table = torch.zeros(n_rows, E)
        idx = torch.randperm(n_rows, generator=g)[:30]          # 30 of 50 rows: most rows are touched by several workers
        self.table[idx] = torch.randn(30, E, generator=g) * (10.0 ** float(rank))     # magnitudes that make the order matter
        self.touched = idx.to(torch.int32)

    def dense(self):
        return self.dense_block.clone()

    def set_dense(self, t):
        self.dense_block = t.clone()

    def export_rows(self):
        return self.touched, self.table[self.touched.long()].clone()

    def add_rows(self, rows, grads):
        self.table.index_add_(0, rows.long(), grads)


def _exchange_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    from dismember_amd import sharding
    from dismember_amd.trainer import exchange_gradients
    dist, r, w, _ = sharding.init_distributed("gloo")
    ports = [_FakePort(torch, k) for k in range(world)]        # every rank can rebuild every rank's local gradient
    mine = ports[r]
    n = exchange_gradients(mine, dist, torch)
    want_dense = sum(p.dense_block for p in [_FakePort(torch, k) for k in range(world)])
    want_table = sum(p.table for p in [_FakePort(torch, k) for k in range(world)])
    q.put((r, n, bool(torch.allclose(mine.dense_block, want_dense)), bool(torch.allclose(mine.table, want_table, rtol=1e-5, atol=1e-5)),
           mine.table.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_gradient_exchange_bit_identical_replicas():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 3, port, q)) for r in range(3)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(3))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(n == 3 and dense_ok and table_ok for _, n, dense_ok, table_ok, _ in out)
    assert out[0][4] == out[1][4] == out[2][4]          # every replica holds the same bits (rows summed in rank order)


# ---- item-sharded JTM child weights (2 gloo ranks; the per-item row function stands in for the GPU scorer) ----
def _jtm_rows(lo, hi):
    i = np.arange(lo, hi, dtype=np.float64)[:, None]
    return (np.sin(i * 0.37 + np.arange(4)[None, :]) * 1000).astype(np.float32)


def _jtm_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dismember_amd import sharding
    dist, r, w, _ = sharding.init_distributed("gloo")
    calls = []

    def compute(lo, hi):
        calls.append((lo, hi))
        return _jtm_rows(lo, hi)

    full = sharding.sharded_rows(compute, 37, dist)
    q.put((r, calls, bool(np.array_equal(full, _jtm_rows(0, 37)))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_item_sharded_rows():
    import torch.multiprocessing as mp
    from dismember_amd import sharding
    assert np.array_equal(sharding.sharded_rows(_jtm_rows, 5, None), _jtm_rows(0, 5))     # single rank: whole range
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_jtm_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert out[0][1] == [(0, 19)] and out[1][1] == [(19, 37)]        # each rank scored only its own items
    assert out[0][2] and out[1][2]                                     # and both hold the bit-identical full matrix
