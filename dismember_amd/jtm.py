"""JTM tree learning: mirror of com.mass.jtm.optim.JTM (jtm/src/main/scala/com/mass/jtm/optim/JTM.scala:8-73).

`JTM(...).optimize()` returns the new projection item id -> leaf code, like the reference's
`Map[Int, Int]`.  The whole loop over the gap steps is ONE library call (dm_jtm_optimize_cached): scoring
(TreeLearning.aggregateWeights), the greedy re-balance of every parent node and the projection stay in HBM.
With `comm` (dismember_amd.comm.Comm: one rank per GPU) the same call is collective and sharded like the
reference's workers (JTM.scala:33-68): item-sharded scoring, the weight slices all-gathered device to device,
the re-balance sharded by parent node once a level has as many parents as ranks — bit-identical to one rank.
Items are iterated in ascending id (the reference iterates a Scala HashMap — only relevant for ties, see
DESIGN.md).
"""
import ctypes as C

import os
import numpy as np

from . import _native as N
from .engine import Engine, _i32, _p


class JTM:
    def __init__(self, engine: Engine, leaf_item_ids, leaf_codes, max_level, item_rows, gap=2, seq_len=10,
                 hierarchical=False, min_level=0, use_mask=True, comm=None):
        """item_rows: dict item id -> int array [n_rows * seq_len] (itemSequenceMap, TreeLearning.scala:34-46)."""
        self.engine = engine
        self.items = np.sort(_i32(leaf_item_ids))
        lut = dict(zip(_i32(leaf_item_ids).tolist(), _i32(leaf_codes).tolist()))
        self.item_code = np.array([lut[int(i)] for i in self.items], np.int32)      # code in the CURRENT tree
        self.max_level, self.gap, self.L = int(max_level), int(gap), int(seq_len)
        self.hierarchical, self.min_level, self.use_mask = bool(hierarchical), int(min_level), bool(use_mask)
        self.comm = comm              # comm.Comm: items sharded over ranks, weight blocks all-gathered (sharding.py)
        off = np.zeros(self.items.size + 1, np.int64)
        rows = []
        for k, it in enumerate(self.items.tolist()):
            r = _i32(item_rows.get(it, np.zeros(0, np.int32))).ravel()
            assert r.size % self.L == 0
            off[k + 1] = off[k] + r.size // self.L
            rows.append(r)
        self.row_off = off
        self.row_ids = _i32(np.concatenate(rows)) if off[-1] > 0 else np.zeros(self.L, np.int32)

    @classmethod
    def from_arrays(cls, engine, items_sorted, item_code, max_level, row_off, row_ids, gap=2, seq_len=10, hierarchical=False,
                    min_level=0, use_mask=True, comm=None):
        """The same object from flat arrays (catalogue-scale runs: no per-item Python objects): items in ascending id order,
        item_code[k] the current leaf code of items_sorted[k], row_off [n+1] / row_ids [rows * seq_len] the CSR of the items'
        training rows (itemSequenceMap)."""
        o = cls.__new__(cls)
        o.engine = engine
        o.items = _i32(items_sorted); o.item_code = _i32(item_code)
        o.max_level, o.gap, o.L = int(max_level), int(gap), int(seq_len)
        o.hierarchical, o.min_level, o.use_mask = bool(hierarchical), int(min_level), bool(use_mask)
        o.comm = comm
        o.row_off = np.ascontiguousarray(row_off, np.int64)
        o.row_ids = _i32(row_ids)
        assert o.row_off.size == o.items.size + 1 and o.row_ids.size >= int(o.row_off[-1]) * o.L
        return o

    def weights_range(self, item_node, old_level, level, lo, hi):
        """TreeLearning.aggregateWeights for the items [lo, hi) of the ascending-id item list (their rows only)."""
        nchild = 1 << (level - old_level)
        node = _i32(item_node)
        n = hi - lo
        w = np.empty((max(n, 1), nchild), np.float32)
        if n == 0:
            return w[:0]
        if getattr(self, "_cached", False):          # rows already on the device (optimize): only the node list goes up
            sub = np.ascontiguousarray(node[lo:hi])
            self.engine._chk(N.lib().dm_jtm_child_weights_cached(self.engine._h, _p(sub, N.i32p), lo, n, old_level, level,
                                                                 int(self.hierarchical), self.min_level, int(self.use_mask), _p(w, N.f32p)))
            return w[:n]
        off = np.ascontiguousarray(self.row_off[lo:hi + 1] - self.row_off[lo])
        r0, r1 = int(self.row_off[lo]), int(self.row_off[hi])
        rows = np.ascontiguousarray(self.row_ids[r0 * self.L:max(r1, r0 + 1) * self.L])
        sub = np.ascontiguousarray(node[lo:hi])
        self.engine._chk(N.lib().dm_jtm_child_weights(self.engine._h, _p(off, N.i64p), _p(rows, N.i32p), _p(sub, N.i32p), n,
                                                      self.L, old_level, level, int(self.hierarchical), self.min_level,
                                                      int(self.use_mask), _p(w, N.f32p)))
        return w[:n]

    def child_weights(self, item_node, old_level, level):
        """All items; with `comm` every rank scores its contiguous item range and the blocks are all-gathered."""
        from .sharding import sharded_rows
        return sharded_rows(lambda lo, hi: self.weights_range(item_node, old_level, level, lo, hi), self.items.size, self.comm)

    def rebalance(self, weights, old_node, node, old_level, level, max_assign):
        weights = np.ascontiguousarray(weights, np.float32)
        old_node = _i32(old_node)
        out = np.empty(old_node.size, np.int32)
        self.engine._chk(N.lib().dm_jtm_rebalance(self.engine._h, _p(weights, N.f32p), _p(old_node, N.i32p), old_node.size,
                                                  int(node), old_level, level, int(max_assign), _p(out, N.i32p)))
        return out

    @staticmethod
    def ancestor_at_level(codes, level):
        """JTMTree.getAncestorAtLevel (JTMTree.scala:36-43), vectorised over codes."""
        c1 = np.asarray(codes, np.int64) + 1                    # heap code + 1 = 1 b_1 b_2 ...: its bit length is level + 1
        lv = np.frexp(c1.astype(np.float64))[1] - 1             # exact below 2^53
        return ((c1 >> np.maximum(lv - level, 0)) - 1).astype(np.int32)

    def optimize(self, weight_fn=None, timing=None, as_array=False):
        """JTM.optimize (JTM.scala:22-73).  weight_fn(item_node, old_level, level) -> [n, 2^gap] overrides the GPU
        scorer (parity tests feed the oracle's weights through the same assignment logic).  timing: dict that receives the
        seconds spent in scoring (dm_jtm_child_weights incl. its copies), re-balance (dm_jtm_rebalance_all) and host glue."""
        import time
        proj = np.zeros(self.items.size, np.int32)            # first all assigned to the root (:23-26)
        t_sc = t_rb = t_host = 0.0
        t0 = time.perf_counter()
        if weight_fn is None and self.row_off[0] == 0:         # itemSequenceMap goes to the device once for all gap steps
            lo, hi = 0, int(self.items.size)
            if self.comm is not None and hasattr(self.comm, "_c") and self.comm.world > 1 and os.environ.get("DM_JTM_FUSED", "1") not in ("0", "step"):
                # a rank of a sharded run only scores its item range (dm_jtm_shard_range): only that range's rows go up
                a_, b_ = C.c_int64(0), C.c_int64(0)
                self.engine._chk(N.lib().dm_jtm_shard_range(self.items.size, self.comm.rank, self.comm.world, C.byref(a_), C.byref(b_)))
                lo, hi = a_.value, b_.value
            self.engine._chk(N.lib().dm_jtm_cache_rows_range(self.engine._h, _p(self.row_off, N.i64p), _p(self.row_ids, N.i32p), self.items.size, self.L, lo, hi))
            self._cached = True
            self._cached_rows = int(self.row_off[hi] - self.row_off[lo])
        t_up = time.perf_counter() - t0
        try:
            lib_comm = self.comm is None or hasattr(self.comm, "_c")     # None, or the library's own communicator (not a test adapter)
            if weight_fn is None and lib_comm and getattr(self, "_cached", False) and os.environ.get("DM_JTM_FUSED", "1") not in ("0", "step"):
                # the whole loop over the gap steps in one call, the projection stays on the device between the steps; with a
                # communicator the call is collective: sharded scoring / re-balance, RCCL all-gathers inside the library
                if self.comm is not None:
                    self.engine.attach_comm(self.comm)
                t1 = time.perf_counter()
                out = np.empty(self.items.size, np.int32)
                secs = (C.c_double * 2)()
                self.engine._chk(N.lib().dm_jtm_optimize_cached(self.engine._h, _p(self.item_code, N.i32p), self.items.size, self.max_level, self.gap,
                                                                int(self.hierarchical), self.min_level, int(self.use_mask), _p(out, N.i32p), secs))
                if timing is not None:
                    timing.update(scoring_s=secs[0], rebalance_s=secs[1], host_glue_s=0.0, rows_upload_s=t_up, rows_uploaded=self._cached_rows, fused_step_s=time.perf_counter() - t1,
                                  sharding=self.optimize_stats())
                return out if as_array else dict(zip(self.items.tolist(), out.tolist()))
            return self._optimize(proj, weight_fn, timing, as_array, t_up)
        finally:
            if getattr(self, "_cached", False):
                self._cached = False
                self.engine._chk(N.lib().dm_jtm_cache_rows(self.engine._h, None, None, 0, self.L))

    def optimize_stats(self):
        """dm_jtm_optimize_stats of the last fused optimize on this engine: ranks, transport, items this rank scored / re-balanced,
        replicated and node-sharded steps, bytes all-gathered, seconds in scoring / re-balance / exchange."""
        o = (C.c_uint64 * 10)(); t = (C.c_double * 3)()
        self.engine._chk(N.lib().dm_jtm_optimize_stats(self.engine._h, o, t))
        tr = {0: "host", 1: "rccl"}.get(int(o[1]) if o[1] < 2 else -1, "none")
        return {"nranks": int(o[0]), "transport": tr, "items_scored": int(o[2]), "items_rebalanced_sharded": int(o[3]),
                "steps_replicated_rebalance": int(o[4]), "steps_node_sharded": int(o[5]), "weight_bytes_gathered": int(o[6]),
                "projection_bytes_gathered": int(o[7]), "scoring_s": t[0], "rebalance_s": t[1], "exchange_s": t[2]}

    def _optimize(self, proj, weight_fn, timing, as_array, t_up):
        import time
        t_sc = t_rb = t_host = t_step = 0.0
        c1 = lv = None
        fused = os.environ.get("DM_JTM_FUSED", "1") != "0"
        for old_level in range(0, self.max_level, self.gap):
            level = min(self.max_level, old_level + self.gap)
            t0 = time.perf_counter()
            if c1 is None:                                     # the items' codes do not change during a run: level of each, once
                c1 = self.item_code.astype(np.int64) + 1
                lv = (np.frexp(c1.astype(np.float64))[1] - 1).astype(np.int64)
            old_node = ((c1 >> np.maximum(lv - level, 0)) - 1).astype(np.int32)     # JTMTree.getAncestorAtLevel
            max_assign = 1 << (self.max_level - level)         # TreeLearning.scala:56
            if weight_fn is None and self.comm is None and getattr(self, "_cached", False) and fused:
                # single rank, rows cached: scoring and re-balance of the step in one call, weights never leave the device
                new = np.empty_like(proj)
                t1 = time.perf_counter()
                self.engine._chk(N.lib().dm_jtm_step_cached(self.engine._h, _p(proj, N.i32p), _p(old_node, N.i32p), proj.size, old_level, level,
                                                            int(self.hierarchical), self.min_level, int(self.use_mask), int(max_assign), _p(new, N.i32p)))
                t_host += t1 - t0; t_step += time.perf_counter() - t1
                a_, b_ = C.c_double(0), C.c_double(0)
                self.engine._chk(N.lib().dm_jtm_last_step_seconds(self.engine._h, C.byref(a_), C.byref(b_)))
                t_sc += a_.value; t_rb += b_.value
                proj = new
                continue
            t_host += time.perf_counter() - t0
            t0 = time.perf_counter()
            w = (weight_fn or self.child_weights)(proj, old_level, level)
            t1 = time.perf_counter()
            w = np.ascontiguousarray(w, np.float32)
            new = np.empty_like(proj)                          # every parent node of the level in one call
            t2 = time.perf_counter()
            self.engine._chk(N.lib().dm_jtm_rebalance_all(self.engine._h, _p(w, N.f32p), _p(_i32(old_node), N.i32p), _p(proj, N.i32p),
                                                          proj.size, old_level, level, int(max_assign), _p(new, N.i32p)))
            t3 = time.perf_counter()
            t_sc += t1 - t0; t_host += t2 - t1; t_rb += t3 - t2
            proj = new
        if timing is not None:
            timing.update(scoring_s=t_sc, rebalance_s=t_rb, host_glue_s=t_host, rows_upload_s=t_up, fused_step_s=t_step)
        if as_array:
            return proj
        return dict(zip(self.items.tolist(), proj.tolist()))
