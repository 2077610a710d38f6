"""Data-parallel training of the DIN scorer: mirror of the reference's LocalOptimizer
(tdm/src/main/scala/com/mass/tdm/optim/LocalOptimizer.scala:58-187) with one worker per GPU.

Reference: N worker threads, each with a model clone sharing the weights and a private gradient buffer;
per iteration every worker runs trainBatch on its slice, syncGradients sums the N buffers and divides by N,
then ONE Adam step updates the shared weights.  Here a worker is a process with its own Engine (replicated
weights); `Engine.train_sync_gradients()` (dm_train_sync_gradients: RCCL all-reduce of the dense block + all-gather of
the touched embedding rows, re-summed in rank order, inside the library) makes every replica hold the same summed
gradient, and every replica applies the same Adam step, so replicas stay bit-identical without a parameter broadcast
(never a ring over the 17 GB table).
"""
import numpy as np


class TDMTrainer:
    """One worker of LocalOptimizer.optimize: sample negatives, forward/backward, exchange, Adam.

    comm: dismember_amd.comm.Comm (or None for a single worker).  sampler: "device" draws the negatives on the GPU
    (dm_tdm_sample_train_batch_dev; rows never visit the host), "host" uses dm_tdm_make_train_batch."""

    def __init__(self, engine, neg_counts, start_level=1, use_mask=True, lr=1e-3, comm=None, seed=0, sampler="device",
                 with_prob=False, tolerance=20):
        self.e, self.neg, self.start, self.use_mask = engine, np.asarray(neg_counts, np.int32), start_level, use_mask
        self.comm, self.seed, self.it, self.sampler, self.with_prob = comm, seed, 0, sampler, bool(with_prob)
        self.tolerance = int(tolerance)                   # model.sample_tolerance (NegativeSampler.scala:116-145)
        self.sync_s, self.sync_calls = 0.0, 0             # wall time spent in the gradient exchange (bench.py reports it)
        engine.train_init(lr=lr)
        if comm is not None:
            engine.attach_comm(comm)

    def step(self, seq_item_ids, target_item_ids):
        """One iteration on this worker's slice of the batch; returns the mean loss over workers."""
        rank = self.comm.rank if self.comm is not None else 0
        world = self.comm.world if self.comm is not None else 1
        seed = self.seed + 1000003 * self.it + rank
        if self.sampler == "device":
            loss = self.e.train_step_sampled(seq_item_ids, target_item_ids, self.neg, self.start, seed=seed,
                                             use_mask=self.use_mask, with_prob=self.with_prob, tolerance=self.tolerance)
        else:
            codes, seqs, mask, y = self.e.make_train_batch(seq_item_ids, target_item_ids, self.neg, self.start, seed=seed,
                                                            use_mask=self.use_mask, with_prob=self.with_prob, tolerance=self.tolerance)
            loss = self.e.train_forward_backward(codes, seqs, self.e.rowmask_to_flat(mask, seqs.shape[1]), y)
        if self.comm is not None:
            import time
            t0 = time.perf_counter()
            self.e.train_sync_gradients()                 # syncGradients, LocalOptimizer.scala:164-187
            self.sync_s += time.perf_counter() - t0; self.sync_calls += 1
        self.e.adam_step(1.0 / world)                     # ... / realParallelism, folded into the Adam kernel
        self.it += 1
        if self.comm is not None:
            loss = self.comm.allreduce(float(loss)) / world           # lossSum / realParallelism, LocalOptimizer.scala:161
        return loss
