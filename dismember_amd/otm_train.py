"""OTM training on the GPU: pseudo targets, fixed-weights beam nodes and one optimizer step per level.

Mirror of otm/src/main/scala/com/mass/otm/optim/LocalOptimizer.scala:55-140 with
OTMTree.optimalPseudoTargets / beamSearchNodes (otm/.../tree/OTMTree.scala:27-91,104-212) and
MiniBatch.batchTransform (otm/.../dataset/MiniBatch.scala:17-40).  Every model evaluation runs on the device
(general-rows forward, beam kernel in OTM mode with its per-level trace, training kernels); the label bookkeeping
is small host code.  The arithmetic follows the loaded model: an f64 model (the reference's DIN[Double]) is searched by
the fp64 beam kernel and trained by the fp64 kernels with fp64 Adam state; an f32 model runs the throughput kernels.

computeTargets' mirrored prediction offsets (a reference quirk, OTMTree.scala:115-128) are reproduced.
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

    def compute_targets(self, children, seqs):
        pos, neg, neg_labels, rows = [], [], [], []
        for u, nodes in enumerate(children):
            ids = [n for n, _ in nodes]
            sib = [n - 1 if n % 2 == 0 else n + 1 for n in ids]
            lut = dict(nodes)
            pos += ids; neg += sib
            neg_labels += [lut.get(s, 0.0) for s in sib]
            rows += [seqs[u]] * len(ids)
        if pos:
            preds = self._forward(pos + neg, rows + rows)          # one launch for both children (OTMTree.scala:159-164)
            pos_preds, neg_preds = preds[:len(pos)], preds[len(pos):]
        else:
            pos_preds = neg_preds = np.zeros(0, np.float32)
        out = [None] * len(children)
        offset = 0
        for u in range(len(children) - 1, -1, -1):                 # foldRight with offset from 0 (reference quirk)
            acc = {}
            for i, (n, score) in enumerate(children[u]):
                idx = offset + i
                label = score if pos_preds[idx] >= neg_preds[idx] else neg_labels[idx]
                par = (n - 1) >> 1
                acc[par] = acc.get(par, 0.0) + label
            out[u] = {k: min(1.0, max(0.0, v)) for k, v in acc.items()}
            offset += len(children[u])
        return out

    def optimal_pseudo_targets(self, target_nodes, seqs):
        cur = [[(int(t), 1.0) for t in tl] for tl in target_nodes]
        levels = [[dict(c) for c in cur]]
        for _ in range(self.leaf_level - 1, self.start_level, -1):
            nxt = self.compute_targets(cur, seqs)
            levels.insert(0, nxt)
            cur = [list(d.items()) for d in nxt]
        return levels

    def train_batch(self, seqs, target_nodes):
        """One LocalOptimizer iteration for a batch of users (seqs: node ids, -1 pad; target_nodes: leaf node ids per
        user).  Returns the per-level losses (one Adam step per level, LocalOptimizer.scala:73-80)."""
        seqs = _i32(seqs)
        targets = self.optimal_pseudo_targets(target_nodes, seqs)
        beams = self.beam_search_nodes(seqs)
        losses = []
        for lv, (tl, bl) in enumerate(zip(targets, beams)):
            codes, rows, labels = [], [], []
            for u, cand in enumerate(bl):
                for n, _ in cand:
                    codes.append(n); rows.append(seqs[u]); labels.append(tl[u].get(n, 0.0))
            rows = _i32(rows).reshape(-1, self.L)
            pad = np.flatnonzero(rows.reshape(-1) == -1).astype(np.int32)
            loss = self.e.train_forward_backward(_i32(codes), rows, pad, np.asarray(labels, np.float32))
            world = 1
            if self.comm is not None:
                world = self.comm.world
                self.e.train_sync_gradients()
                loss = self.comm.allreduce(float(loss)) / world
            losses.append(loss)
            self.e.adam_step(1.0 / world)
        return losses
