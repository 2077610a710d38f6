"""The gradient exchange behind the C ABI (dm_train_sync_gradients = LocalOptimizer.syncGradients,
tdm/src/main/scala/com/mass/tdm/optim/LocalOptimizer.scala:164-187) on real engines.

A GPU box has ONE device and RCCL refuses two ranks per device, so the multi-worker protocol runs as W processes sharing
the GPU over the library's host transport (same entry point, same kernels: export, zero, rank-ordered re-summation; only
the wire differs), and RCCL itself is exercised as a single-rank communicator (ncclCommInitRank, ncclBroadcast path of
dm_comm_all_gather_dev)."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batch(rank, B=96, L=10, NI=8191):
    rng = np.random.default_rng(500 + rank)
    codes = rng.integers(1, NI, B).astype(np.int32)
    seqs = rng.integers(0, 400, (B, L)).astype(np.int32)        # a small history vocabulary: most rows are touched by every worker
    seqs[rng.random((B, L)) < 0.2] = -1
    labels = (rng.random(B) < 0.3).astype(np.float32)
    return codes, seqs, np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32), labels


def _worker(rank, world, port, q, scale):
    from dismember_amd import Engine
    from dismember_amd.comm import Comm
    w = np.load(os.path.join(GOLDEN, "din_f32.npy"))
    eng = Engine(0)
    eng.load_weights_din(w, 16, 8191)
    eng.train_init(lr=1e-3)
    comm = Comm(world, rank, "127.0.0.1", port, transport="host")
    eng.attach_comm(comm)
    codes, seqs, pad, y = _batch(rank)
    y = y * np.float32(scale ** rank)           # very different gradient magnitudes per worker: the summation order is visible
    loss = eng.train_forward_backward(codes, seqs, pad, y)
    local = eng.train_download("grad")
    eng.train_sync_gradients()
    summed = eng.train_download("grad")
    eng.adam_step(1.0 / world)
    wts = eng.train_download("weights")
    gathered = eng.comm_all_gather_dev(np.full((rank + 2, 3), rank, np.int32))
    q.put((rank, loss, local, summed, wts, gathered))
    comm.barrier()
    eng.close(); comm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sync_gradients_replicas_bit_identical(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 37.0)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = np.zeros_like(out[0][2])
    for r in range(world):
        want = want + out[r][2]                 # float32, rank order: 0 + g_0 + g_1 + ... exactly what every replica forms
    touched = np.flatnonzero(np.abs(want[:8191 * 16]).reshape(8191, 16).sum(1) > 0)
    assert touched.size > 300
    for r in range(world):
        assert np.array_equal(out[r][3], want), r                       # the sum, bit for bit, on every replica
        assert np.array_equal(out[r][4], out[0][4])                      # and identical weights after the Adam step
        assert np.array_equal(out[r][5], np.concatenate([np.full((k + 2, 3), k, np.int32) for k in range(world)]))
    assert not np.array_equal(out[0][4], np.load(os.path.join(GOLDEN, "din_f32.npy")))


def test_rccl_single_rank_communicator():
    from dismember_amd import Engine
    from dismember_amd.comm import Comm
    eng = Engine(0)
    eng.load_weights_din(np.load(os.path.join(GOLDEN, "din_f32.npy")), 16, 8191)
    eng.train_init(lr=1e-3)
    comm = Comm(1, 0, "127.0.0.1", _free_port(), transport="rccl", device_id=0)       # ncclCommInitRank on the real device
    eng.attach_comm(comm)
    codes, seqs, pad, y = _batch(0)
    eng.train_forward_backward(codes, seqs, pad, y)
    g0 = eng.train_download("grad")
    eng.train_sync_gradients()                                                        # one worker: the gradient is its own
    assert np.array_equal(eng.train_download("grad"), g0)
    a = np.arange(24, dtype=np.float32).reshape(8, 3)
    assert np.array_equal(eng.comm_all_gather_dev(a), a)                              # ncclBroadcast through RCCL
    assert comm.allreduce([1.5, 2.0]).tolist() == [1.5, 2.0] and comm.all_gather_bytes(b"abc") == [b"abc"]
    comm.barrier()
    eng.close(); comm.close()


def test_sync_without_communicator_fails():
    from dismember_amd import DismemberError, Engine
    eng = Engine(0)
    eng.load_weights_din(np.load(os.path.join(GOLDEN, "din_f32.npy")), 16, 8191)
    eng.train_init()
    with pytest.raises(DismemberError) as e:
        eng.train_sync_gradients()
    assert e.value.code == -3
    eng.close()


def _otm_worker(rank, world, port, q):
    from dismember_amd import Engine
    from dismember_amd.comm import Comm
    from dismember_amd.otm_train import OTMTrainer
    w = np.load(os.path.join(GOLDEN, "din_f32.npy"))
    eng = Engine(0)
    eng.load_weights_din(w, 16, 8191)
    comm = Comm(world, rank, "127.0.0.1", port, transport="host")
    tr = OTMTrainer(eng, leaf_level=12, beam=20, lr=1e-3, comm=comm)
    rng = np.random.default_rng(40 + rank)                     # every worker: its own users (node-id histories, leaf targets)
    codes = rng.integers(4095, 8191, (6, 10)).astype(np.int32)
    codes[rng.random((6, 10)) < 0.2] = -1
    targets = [rng.integers(4095, 8191, int(rng.integers(1, 4))).tolist() for _ in range(6)]
    losses = tr.train_batch(codes, targets)
    q.put((rank, losses, eng.train_download("weights")))
    comm.barrier()
    eng.close(); comm.close()


def test_otm_trainer_replicas_stay_identical():
    """BASELINE configs[2]: OTM training with users sharded over workers — one gradient exchange and one Adam step per tree level
    (otm/.../optim/LocalOptimizer.scala:73-80, 217-233); replicas must hold the same bits after a whole iteration."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_otm_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert out[0][1] == out[1][1] and len(out[0][1]) == 12 - 4 and all(np.isfinite(out[0][1]))       # the averaged per-level losses
    assert np.array_equal(out[0][2], out[1][2])
    assert not np.array_equal(out[0][2], np.load(os.path.join(GOLDEN, "din_f32.npy")))


def _clique_train(devices):
    """dm_comm_create_all + dm_allreduce_grads: ONE process drives every GPU of the list (the reference's one-JVM, N-worker shape)."""
    import ctypes as C
    from dismember_amd import Engine, _native as N
    from dismember_amd.comm import make_clique
    w = np.load(os.path.join(GOLDEN, "din_f32.npy"))
    comms = make_clique(devices)
    engs = []
    for i, d in enumerate(devices):
        e = Engine(d); e.load_weights_din(w, 16, 8191); e.train_init(lr=1e-3); e.attach_comm(comms[i]); engs.append(e)
    local = []
    for i, e in enumerate(engs):
        codes, seqs, pad, y = _batch(i)
        e.train_forward_backward(codes, seqs, pad, y)
        local.append(e.train_download("grad"))
    hs = (C.c_void_p * len(engs))(*[e._h for e in engs])
    assert N.lib().dm_allreduce_grads(hs, len(engs)) == 0
    summed = [e.train_download("grad") for e in engs]
    st = engs[0].train_sync_stats()
    for e in engs:
        e.adam_step(1.0 / len(engs))
    wts = [e.train_download("weights") for e in engs]
    for e in engs:
        e.close()
    for c in comms:
        N.lib().dm_comm_destroy(c)
    return local, summed, wts, st


def test_clique_single_device_is_a_no_op():
    """n == 1: dm_comm_create_all gives a one-rank RCCL communicator and dm_allreduce_grads leaves the gradient untouched."""
    local, summed, wts, st = _clique_train([0])
    assert np.array_equal(local[0], summed[0]) and st["nranks"] == 1


def test_clique_all_devices_replicas_identical():
    """Every GPU of the box in one process (skipped on one-GPU boxes): the replicas' summed gradients and updated weights are
    bit-identical, touched embedding rows equal the rank-ordered sum, the exchange needed two host synchronisations."""
    from dismember_amd import _native as N
    import ctypes as C
    n = C.c_int(0)
    N.lib().dm_device_count(C.byref(n))
    if n.value < 2:
        pytest.skip("needs >= 2 GPUs in one process (dm_comm_create_all); this box has %d" % n.value)
    devs = list(range(min(n.value, 4)))
    local, summed, wts, st = _clique_train(devs)
    for r in range(1, len(devs)):
        assert np.array_equal(summed[r], summed[0]) and np.array_equal(wts[r], wts[0])
    NI, E = 8191, 16
    ref = np.zeros_like(local[0][:NI * E])
    for g in local:
        ref = ref + g[:NI * E]                                  # rank order
    assert np.array_equal(summed[0][:NI * E], ref)
    dense = np.sum([g[NI * E:].astype(np.float64) for g in local], axis=0)
    assert np.allclose(summed[0][NI * E:], dense, rtol=1e-5, atol=1e-7)
    assert st["nranks"] == len(devs) and st["transport"] == "rccl" and st["host_syncs"] == 2
