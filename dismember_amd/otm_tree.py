"""OTM tree construction: mirror of `com.mass.otm.tree.TreeConstruction`
(otm/src/main/scala/com/mass/otm/tree/TreeConstruction.scala:18-141) — the JTM re-assignment in OTM's setting: item
sequences hold node ids (-1 = padding), the tree is the complete tree over `upperLog2(#items)` levels, the model and all
sums are fp64.  Scoring runs on the GPU (`dm_otm_child_weights`: every (row, chain node) pair is one row of the DIN
forward in the loaded dtype), the greedy re-balance is `dm_otm_rebalance` (exact host logic).  Items are visited in
ascending id (the reference iterates hash maps; only ties can see the difference)."""
import ctypes as C

import numpy as np

from . import _native as N
from .engine import Engine, _i32, _p
from .sharding import sharded_rows


class TreeConstruction:
    def __init__(self, engine: Engine, item_id_mapping, item_sequences, gap=2, seq_len=10, use_mask=True, comm=None):
        """item_id_mapping: item -> leaf node id; item_sequences: item -> flat [rows * seq_len] NODE ids
        (itemSequenceMap, TreeConstruction.scala:36-43: histories already mapped through itemIdMapping)."""
        self.engine = engine
        self.items = np.array(sorted(int(i) for i in item_id_mapping), np.int32)
        self.item_leaf = np.array([item_id_mapping[int(i)] for i in self.items], np.int32)
        self.leaf_level = int(np.ceil(np.log(len(self.items)) / np.log(2)))          # upperLog2, otm/package.scala:16
        self.gap, self.L, self.use_mask, self.comm = int(gap), int(seq_len), bool(use_mask), comm
        off = np.zeros(self.items.size + 1, np.int64)
        rows = []
        for k, it in enumerate(self.items.tolist()):
            r = _i32(item_sequences.get(it, np.zeros(0, np.int32))).ravel()
            assert r.size % self.L == 0
            off[k + 1] = off[k] + r.size // self.L
            rows.append(r)
        self.row_off = off
        self.row_codes = _i32(np.concatenate(rows)) if off[-1] > 0 else np.full(self.L, -1, np.int32)

    def weights_range(self, item_node, old_level, level, lo, hi):
        nchild = 1 << (level - old_level)
        n = hi - lo
        w = np.empty((max(n, 1), nchild), np.float64)
        if n == 0:
            return w[:0]
        off = np.ascontiguousarray(self.row_off[lo:hi + 1] - self.row_off[lo])
        r0, r1 = int(self.row_off[lo]), int(self.row_off[hi])
        rows = np.ascontiguousarray(self.row_codes[r0 * self.L:max(r1, r0 + 1) * self.L])
        sub = np.ascontiguousarray(_i32(item_node)[lo:hi])
        self.engine._chk(N.lib().dm_otm_child_weights(self.engine._h, _p(off, N.i64p), _p(rows, N.i32p), _p(sub, N.i32p), n, self.L,
                                                      old_level, level, int(self.use_mask), w.ctypes.data_as(C.POINTER(C.c_double))))
        return w[:n]

    def child_weights(self, item_node, old_level, level):
        return sharded_rows(lambda lo, hi: self.weights_range(item_node, old_level, level, lo, hi), self.items.size, self.comm)

    def rebalance(self, weights, old_node, node, old_level, level, max_assign):
        weights = np.ascontiguousarray(weights, np.float64)
        old_node = _i32(old_node)
        out = np.empty(old_node.size, np.int32)
        self.engine._chk(N.lib().dm_otm_rebalance(self.engine._h, weights.ctypes.data_as(C.POINTER(C.c_double)), _p(old_node, N.i32p),
                                                  old_node.size, int(node), old_level, level, int(max_assign), _p(out, N.i32p)))
        return out

    @staticmethod
    def ancestor_at_level(nodes, level):
        c = np.asarray(nodes, np.int64).copy()
        lim = (1 << (level + 1)) - 1
        while True:
            m = c >= lim
            if not m.any():
                return c.astype(np.int32)
            c[m] = (c[m] - 1) >> 1

    def run(self, weight_fn=None):
        """TreeConstruction.run (:44-101): item -> new leaf node."""
        proj = np.zeros(self.items.size, np.int32)
        for old_level in range(0, self.leaf_level, self.gap):
            level = min(self.leaf_level, old_level + self.gap)
            w = (weight_fn or self.child_weights)(proj, old_level, level)
            old_node = self.ancestor_at_level(self.item_leaf, level)
            max_assign = 1 << (self.leaf_level - level)
            w = np.ascontiguousarray(w, np.float64)
            new = np.empty_like(proj)
            self.engine._chk(N.lib().dm_otm_rebalance_all(self.engine._h, w.ctypes.data_as(C.POINTER(C.c_double)), _p(_i32(old_node), N.i32p),
                                                          _p(proj, N.i32p), proj.size, old_level, level, int(max_assign), _p(new, N.i32p)))
            proj = new
        return dict(zip(self.items.tolist(), proj.tolist()))
