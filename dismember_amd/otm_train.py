"""OTM training on the GPU: one LocalOptimizer iteration as ONE library call (dm_otm_train_batch).

Mirror of otm/src/main/scala/com/mass/otm/optim/LocalOptimizer.scala:55-140 with
OTMTree.optimalPseudoTargets / beamSearchNodes (otm/.../tree/OTMTree.scala:27-91,104-212) and
MiniBatch.batchTransform (otm/.../dataset/MiniBatch.scala:17-40).  Pseudo targets (sibling pairs, the mirrored prediction
offsets of computeTargets — a reference quirk, OTMTree.scala:115-128 — and clip(sum of children)), the beam nodes and the
per-level label join all run on the device inside the library (csrc/otm_train.hip.inc); this class only flattens the batch.
The arithmetic follows the loaded model: an f64 model (the reference's DIN[Double]) is searched by the fp64 beam kernel and
trained by the fp64 kernels with fp64 Adam state; an f32 model runs the throughput kernels.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .engine import _i32, _p


def lower_log2(n):
    return int(n).bit_length() - 1


class OTMTrainer:
    def __init__(self, engine, leaf_level, beam, seq_len=10, lr=1e-3, comm=None):
        """comm: dismember_amd.comm.Comm — one worker per GPU, users sharded over the workers; every level's gradients go
        through dm_train_sync_gradients (syncGradients, otm/.../optim/LocalOptimizer.scala:217-233) before the shared Adam step."""
        self.e, self.leaf_level, self.beam, self.L = engine, int(leaf_level), int(beam), int(seq_len)
        self.start_level = lower_log2(beam)
        self.comm = comm
        engine.train_init(lr=lr)
        if comm is not None:
            engine.attach_comm(comm)

    # ---- model evaluations
    def _forward(self, nodes, row_seqs):
        seqs = _i32(row_seqs).reshape(-1, self.L)
        pad = np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32)
        return self.e.din_forward(_i32(nodes), seqs, pad)

    def beam_search_nodes(self, seqs):
        """Per level (start+1 .. leaf) and user the scored candidate list, weights fixed (OTMTree.scala:67-91)."""
        seqs = _i32(seqs)
        U = seqs.shape[0]
        levels = self.leaf_level - self.start_level
        cap = max(32, ((2 * self.beam + 15) // 16) * 16)
        if self.e.dtype == np.float64 and self.e.scorer_mode()["mode"] == "f64":
            _, _, _, tc, ts, tn = self.e.otm_beam_search_f64(seqs, self.beam, self.leaf_level, trace_levels=levels)
        else:
            ids = np.empty((U, 2 * self.beam), np.int32); sc = np.empty((U, 2 * self.beam), np.float32); cnt = np.empty(U, np.int32)
            tc = np.zeros((U, levels, cap), np.int32); ts = np.zeros((U, levels, cap), np.float32); tn = np.zeros((U, levels), np.int32)
            self.e._chk(N.lib().dm_otm_beam_search_trace(self.e._h, _p(seqs, N.i32p), U, self.L, self.beam, self.leaf_level,
                                                         _p(ids, N.i32p), _p(sc, N.f32p), _p(cnt, N.i32p), levels,
                                                         _p(tc, N.i32p), _p(ts, N.f32p), _p(tn, N.i32p)))
        return [[list(zip(tc[u, lv, :tn[u, lv]].tolist(), ts[u, lv, :tn[u, lv]].tolist())) for u in range(U)]
                for lv in range(levels)]

    def _flatten(self, seqs, target_nodes):
        seqs = _i32(seqs).reshape(-1, self.L)
        off = np.zeros(seqs.shape[0] + 1, np.int64)
        off[1:] = np.cumsum([len(t) for t in target_nodes])
        flat = _i32(np.concatenate([np.asarray(t, np.int32).ravel() for t in target_nodes])) if off[-1] > 0 else np.zeros(1, np.int32)
        return seqs, off, flat

    def _opts(self, target_mode):
        return N.OtmTrainOpts(self.beam, self.leaf_level, 1, {"pseudo": 0, "normal": 1}[target_mode])

    def optimal_pseudo_targets(self, target_nodes, seqs, target_mode="pseudo"):
        """OTMTree.optimalPseudoTargets (or normalTargets) on the device (dm_otm_pseudo_targets): list over the levels
        start+1 .. leaf of per-user {node: label} in the reference's list order (first appearance)."""
        seqs, off, flat = self._flatten(seqs, target_nodes)
        U, levels, NT = seqs.shape[0], self.leaf_level - self.start_level, max(int(off[-1]), 1)
        nodes = np.empty((levels, NT), np.int32); labels = np.empty((levels, NT), np.float64); counts = np.empty((levels, U), np.int32)
        opts = self._opts(target_mode)
        self.e._chk(N.lib().dm_otm_pseudo_targets(self.e._h, _p(seqs, N.i32p), U, self.L, _p(off, N.i64p), _p(flat, N.i32p), C.byref(opts),
                                                  _p(nodes, N.i32p), labels.ctypes.data_as(C.POINTER(C.c_double)), _p(counts, N.i32p)))
        out = []
        for lv in range(levels):
            per = []
            for u in range(U):
                b, n = int(off[u]), int(counts[lv, u])
                d = {}
                for k, v in zip(nodes[lv, b:b + n].tolist(), labels[lv, b:b + n].tolist()):
                    d.setdefault(k, v)              # normal mode may repeat an ancestor: List.find sees the first
                per.append(d)
            out.append(per)
        return out

    def train_batch(self, seqs, target_nodes, target_mode="pseudo"):
        """One LocalOptimizer iteration for a batch of users (seqs: node ids, -1 pad; target_nodes: leaf node ids per
        user) in one library call.  Returns the per-level losses, averaged over the workers when a communicator is attached
        (one gradient exchange and one Adam step per level, LocalOptimizer.scala:73-80)."""
        seqs, off, flat = self._flatten(seqs, target_nodes)
        levels = self.leaf_level - self.start_level
        losses = (C.c_double * levels)()
        nl = C.c_int(0)
        opts = self._opts(target_mode)
        self.e._chk(N.lib().dm_otm_train_batch(self.e._h, _p(seqs, N.i32p), seqs.shape[0], self.L, _p(off, N.i64p), _p(flat, N.i32p),
                                               C.byref(opts), losses, C.byref(nl)))
        return [float(losses[i]) for i in range(nl.value)]

    def last_stats(self):
        """dm_otm_train_stats: users, levels, rows of the pseudo-target forwards, rows trained; seconds per phase."""
        o = (C.c_uint64 * 6)(); t = (C.c_double * 5)()
        self.e._chk(N.lib().dm_otm_train_stats(self.e._h, o, t))
        return {"users": int(o[0]), "levels": int(o[1]), "pseudo_target_forward_rows": int(o[2]), "rows_trained": int(o[3]),
                "pseudo_targets_s": t[0], "beam_search_s": t[1], "forward_backward_s": t[2], "exchange_s": t[3], "adam_s": t[4]}
