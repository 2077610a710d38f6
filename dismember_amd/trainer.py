"""Data-parallel training of the DIN scorer: mirror of the reference's LocalOptimizer
(tdm/src/main/scala/com/mass/tdm/optim/LocalOptimizer.scala:58-187) with one worker per GPU.

Reference: N worker threads, each with a model clone sharing the weights and a private gradient buffer;
per iteration every worker runs trainBatch on its slice, syncGradients sums the N buffers and divides by N,
then ONE Adam step updates the shared weights.  Here a worker is a process with its own Engine (replicated
weights); `exchange_gradients` makes every replica hold the same summed gradient, and every replica applies
the same Adam step, so replicas stay bit-identical without a parameter broadcast:
  * dense block (att.W, l1.W, l1.b, l2.W, l2.b; 198 KB at E=128): all-reduce(sum)
  * embedding gradients: row-sparse -> all-gather of (row index, gradient row) lists; every rank then rebuilds each touched
    row as 0 + g_0 + g_1 + ... in rank order (its own contribution is subtracted first), so the sums are bit-identical on
    all replicas for any number of workers (never a ring over the 17 GB table).
"""
import ctypes as C

import numpy as np

from . import _native as N


class EngineGradPort:
    """Adapter: the gradient-exchange protocol over a real Engine, buffers in torch device memory."""

    def __init__(self, engine, torch):
        self.e, self.torch = engine, torch

    def dense(self):
        p, n = C.c_void_p(), C.c_int64()
        self.e._chk(N.lib().dm_train_dense_block(self.e._h, C.byref(p), C.byref(n)))
        t = self.torch.empty(n.value, dtype=self.torch.float32, device="cuda")
        self.e._chk(N.lib().dm_memcpy_d2d(self.e._h, C.c_void_p(t.data_ptr()), p, n.value * 4))
        self._dense_ptr = p
        return t

    def set_dense(self, t):
        self.e._chk(N.lib().dm_memcpy_d2d(self.e._h, self._dense_ptr, C.c_void_p(t.data_ptr()), t.numel() * 4))

    def export_rows(self):
        n = C.c_int64()
        self.e._chk(N.lib().dm_train_export_rows(self.e._h, None, None, 0, C.byref(n)))
        rows = self.torch.empty(max(n.value, 1), dtype=self.torch.int32, device="cuda")
        grads = self.torch.empty((max(n.value, 1), self.e.E), dtype=self.torch.float32, device="cuda")
        self.e._chk(N.lib().dm_train_export_rows(self.e._h, C.c_void_p(rows.data_ptr()), C.c_void_p(grads.data_ptr()),
                                                 rows.numel(), C.byref(n)))
        return rows[:n.value], grads[:n.value]

    def add_rows(self, rows, grads):
        rows, grads = rows.contiguous(), grads.contiguous()
        self.e._chk(N.lib().dm_train_add_rows(self.e._h, C.c_void_p(rows.data_ptr()), C.c_void_p(grads.data_ptr()),
                                              rows.numel()))


def exchange_gradients(port, dist, torch):
    """syncGradients (LocalOptimizer.scala:164-187) minus the final division (folded into the Adam step).
    `port` exposes dense()/set_dense()/export_rows()/add_rows(); dist=None is the single-worker case."""
    if dist is None:
        return 1
    world, rank = dist.get_world_size(), dist.get_rank()
    d = port.dense()
    dist.all_reduce(d, op=dist.ReduceOp.SUM)
    port.set_dense(d)
    rows, grads = port.export_rows()
    n_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    cap = max(int(s.item()) for s in sizes)
    if cap > 0:
        prow = torch.full((cap,), -1, dtype=rows.dtype, device=rows.device)
        pgrad = torch.zeros((cap, grads.shape[1]), dtype=grads.dtype, device=grads.device)
        prow[:rows.shape[0]] = rows
        pgrad[:rows.shape[0]] = grads
        all_rows = [torch.empty_like(prow) for _ in range(world)]
        all_grads = [torch.empty_like(pgrad) for _ in range(world)]
        dist.all_gather(all_rows, prow)
        dist.all_gather(all_grads, pgrad)
        # Every replica must form each row's sum in the SAME order, or rows touched by three or more workers differ in
        # the last bit between replicas ((g2 + g0) + g1 != (g0 + g1) + g2 in floating point).  So the rank's own
        # contribution is taken out again (x + (-x) is exactly 0) and all contributions are added in rank order 0..W-1.
        if rows.shape[0] > 0:
            port.add_rows(rows, -grads)
        for r in range(world):
            k = int(sizes[r].item())
            if k > 0:
                port.add_rows(all_rows[r][:k], all_grads[r][:k])
    return world


class TDMTrainer:
    """One worker of LocalOptimizer.optimize: sample negatives, forward/backward, exchange, Adam."""

    def __init__(self, engine, neg_counts, start_level=1, use_mask=True, lr=1e-3, dist=None, torch=None, seed=0):
        self.e, self.neg, self.start, self.use_mask = engine, np.asarray(neg_counts, np.int32), start_level, use_mask
        self.dist, self.torch, self.seed, self.it = dist, torch, seed, 0
        engine.train_init(lr=lr)
        self.port = EngineGradPort(engine, torch) if dist is not None else None

    def step(self, seq_item_ids, target_item_ids):
        """One iteration on this worker's slice of the batch; returns the mean loss over workers."""
        rank = self.dist.get_rank() if self.dist is not None else 0
        codes, seqs, mask, y = self.e.make_train_batch(seq_item_ids, target_item_ids, self.neg, self.start,
                                                        seed=self.seed + 1000003 * self.it + rank, use_mask=self.use_mask)
        loss = self.e.train_forward_backward(codes, seqs, self.e.rowmask_to_flat(mask, seqs.shape[1]), y)
        world = exchange_gradients(self.port, self.dist, self.torch)
        self.e.adam_step(1.0 / world)
        self.it += 1
        if self.dist is not None:
            t = self.torch.tensor([loss], dtype=self.torch.float64, device="cuda" if self.dist.get_backend() == "nccl" else "cpu")
            self.dist.all_reduce(t)
            loss = float(t.item()) / world           # lossSum / realParallelism, LocalOptimizer.scala:161
        return loss
