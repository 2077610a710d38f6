"""N>1 path on CPU.  Two kinds of worker processes:
  * the library's own HOST transport (dm_comm_create_tcp, no GPU needed for the host-buffer collectives): what
    bench.py --gpus N and the trainers use, with RCCL as the transport on GPUs;
  * a gloo adapter with the same interface (world_size 2), which pins the sharding helpers against torch.distributed.
The device half of the exchange (dm_train_sync_gradients) is covered on the GPU by tests/test_gpu_comm.py."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from dismember_amd.sharding import shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(target, world, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return out


def test_shard_range_partitions_contiguously():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- item-sharded JTM child weights (the per-item row function stands in for the GPU scorer) ----
def _jtm_rows(lo, hi):
    i = np.arange(lo, hi, dtype=np.float64)[:, None]
    return (np.sin(i * 0.37 + np.arange(4)[None, :]) * 1000).astype(np.float32)


def _serve_and_jtm(comm, q):
    from dismember_amd import sharding
    r, w = comm.rank, comm.world
    n_users = 101
    lo, hi = sharding.shard_range(n_users, r, w)
    # stand-in for the per-rank beam search: "ids" derived from the global user index
    local = np.stack([np.arange(lo, hi) * 10 + k for k in range(3)], axis=1).astype(np.int32)
    comm.barrier()
    slowest = sharding.max_over_ranks(1.0 + r, comm)
    allids = sharding.gather_results(local, comm)
    calls = []

    def compute(a, b):
        calls.append((a, b))
        return _jtm_rows(a, b)

    full = sharding.sharded_rows(compute, 37, comm)
    total = comm.allreduce([float(r + 1), 0.5])
    q.put((r, slowest, allids.shape, bool((allids[:, 0] == np.arange(n_users) * 10).all()), calls,
           bool(np.array_equal(full, _jtm_rows(0, 37))), total.tolist()))
    comm.barrier()


def _host_worker(rank, world, port, q):
    from dismember_amd.comm import Comm
    comm = Comm(world, rank, "127.0.0.1", port, transport="host")
    _serve_and_jtm(comm, q)
    # ragged payloads (rank r sends r * 1000 + 3 bytes; rank 0's may be empty elsewhere in the protocol)
    blocks = comm.all_gather_bytes(bytes([rank + 1]) * (rank * 1000 + 3))
    assert [len(b) for b in blocks] == [r * 1000 + 3 for r in range(world)] and all(set(b) == {r + 1} for r, b in enumerate(blocks))
    assert comm.all_gather_bytes(b"") == [b""] * world
    comm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_host_transport_sharding(world):
    out = _run(_host_worker, world)
    for r, slowest, shape, ordered, calls, same, total in out:
        assert slowest == float(world) and shape == (101, 3) and ordered
        assert calls == [shard_range(37, r, world)] and same      # each rank scored only its own items; bit-identical full matrix
        assert total == [world * (world + 1) / 2, 0.5 * world]


class GlooComm:
    """torch.distributed (gloo) behind the comm interface the sharding helpers use — test infrastructure only."""

    def __init__(self, dist):
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def allreduce(self, values, op="sum"):
        import torch
        t = torch.tensor(np.atleast_1d(np.asarray(values, np.float64)))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        return t.numpy() if np.ndim(values) else float(t[0])

    def all_gather_array(self, arr):
        objs = [None] * self.world
        self.dist.all_gather_object(objs, np.ascontiguousarray(arr))
        return np.concatenate(objs, axis=0)


def _gloo_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo")
    _serve_and_jtm(GlooComm(dist), q)
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    out = _run(_gloo_worker, 2)
    for r, slowest, shape, ordered, calls, same, total in out:
        assert slowest == 2.0 and shape == (101, 3) and ordered and same and calls == [shard_range(37, r, 2)]


def test_comm_create_errors():
    from dismember_amd.comm import Comm, CommError
    with pytest.raises(CommError):
        Comm(2, 5, "127.0.0.1", 1234, transport="host")          # rank outside [0, nranks)
    c = Comm(1, 0, "127.0.0.1", _free_port(), transport="host")   # a single rank needs no peer
    assert c.allreduce(3.0, "max") == 3.0 and c.all_gather_bytes(b"xy") == [b"xy"]
    c.close()
