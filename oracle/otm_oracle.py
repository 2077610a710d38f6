"""TEST INFRASTRUCTURE: restatement of OTM's pseudo-target construction and per-level training step
(otm/src/main/scala/com/mass/otm/tree/OTMTree.scala:27-63,104-212, otm/.../optim/LocalOptimizer.scala:55-140,
otm/.../dataset/MiniBatch.scala:13-51) on top of the C DIN restatement (f64, as the reference runs OTM).
Pure-Python control flow: only for the small cases the parity tests use.

Reference quirk kept on purpose: computeTargets walks the users with foldRight while `offset`
counts from 0, so user k of U reads the predictions stored for the MIRRORED positions (OTMTree.scala:115-128).
"""
import numpy as np

from . import pyoracle as po


def lower_log2(n):
    return po.lib().orc_lower_log2(int(n))


def _forward(din, nodes, seqs_per_row, L):
    """computePreds (OTMTree.scala:167-172): history replicated per row, mask = every -1 position."""
    nodes = np.asarray(nodes, np.int32)
    seqs = np.asarray(seqs_per_row, np.int32).reshape(-1, L)
    pad = np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32)
    return din.forward(nodes, seqs, pad)


def compute_targets(din, children, seqs, L, pred_fn=None):
    """OTMTree.computeTargets (:104-129).  children: per user list of (node id, score).  Returns per user the
    parent-level list of (id, label) — as a dict (the reference's list order is a HashMap's; only membership and
    labels matter downstream)."""
    pos, neg, neg_labels, row_seq = [], [], [], []
    for u, nodes in enumerate(children):
        ids = [n for n, _ in nodes]
        sib = [n - 1 if n % 2 == 0 else n + 1 for n in ids]
        lut = dict(nodes)
        pos += ids
        neg += sib
        neg_labels += [lut.get(s, 0.0) for s in sib]          # nodes.find(_.id == nn) ... else 0.0   (:145-150)
        row_seq += [seqs[u]] * len(ids)
    fwd = pred_fn or (lambda n, s: _forward(din, n, s, L))
    pos_preds = fwd(pos, row_seq) if pos else np.zeros(0)
    neg_preds = fwd(neg, row_seq) if neg else np.zeros(0)
    out = [None] * len(children)
    offset = 0
    for u in range(len(children) - 1, -1, -1):                 # foldRight: last user first, offset from 0
        acc = {}
        for i, (n, score) in enumerate(children[u]):
            idx = offset + i
            label = score if pos_preds[idx] >= neg_preds[idx] else neg_labels[idx]
            par = (n - 1) >> 1
            acc[par] = acc.get(par, 0.0) + label                # groupMapReduce(parent)(_._2)(_ + _)
        out[u] = {k: min(1.0, max(0.0, v)) for k, v in acc.items()}   # clipValue
        offset += len(children[u])
    return out


def optimal_pseudo_targets(din, target_items, seqs, L, start_level, leaf_level, pred_fn=None):
    """OTMTree.optimalPseudoTargets (:27-46): list over levels start+1 .. leaf of per-user {node: label}."""
    levels = []
    cur = [[(int(t), 1.0) for t in tl] for tl in target_items]
    levels.append([dict(c) for c in cur])
    for _ in range(leaf_level - 1, start_level, -1):
        nxt = compute_targets(din, cur, seqs, L, pred_fn)
        levels.insert(0, nxt)
        cur = [list(d.items()) for d in nxt]
    return levels


def beam_search_nodes(din, seqs, L, start_level, leaf_level, beam):
    """OTMTree.beamSearchNodes (:67-91): every level's scored candidates [(id, score)] per user, fixed weights."""
    start = (1 << start_level) - 1
    out = []
    cand = [[(start + i, 0.0) for i in range(start + 1)] for _ in seqs]
    for level in range(start_level, leaf_level):
        nodes = []
        for u, c in enumerate(cand):
            if level == start_level:
                ids = [x for n, _ in c for x in (2 * n + 1, 2 * n + 2)]
            else:
                sc = np.array([s for _, s in c], np.float64)
                order = np.empty(len(c), np.int32)
                po.lib().orc_stable_argsort_desc_f64(sc.ctypes.data_as(po.f64p), order.ctypes.data_as(po.i32p), len(c))
                ids = [x for k in order[:beam] for x in (2 * c[k][0] + 1, 2 * c[k][0] + 2)]
            nodes.append(ids)
        flat = [n for ids in nodes for n in ids]
        rows = [seqs[u] for u, ids in enumerate(nodes) for _ in ids]
        preds = _forward(din, flat, rows, L)
        cand, o = [], 0
        for ids in nodes:
            cand.append(list(zip(ids, preds[o:o + len(ids)].tolist())))
            o += len(ids)
        out.append(cand)
    return out


def level_batch(beam_nodes_level, targets_level, seqs, L):
    """MiniBatch.batchTransform (MiniBatch.scala:17-40): rows = every user's candidates of the level, label = the
    pseudo-target score when the node is a target, else 0."""
    codes, rows, labels = [], [], []
    for u, cand in enumerate(beam_nodes_level):
        for n, _ in cand:
            codes.append(n); rows.append(seqs[u]); labels.append(targets_level[u].get(n, 0.0))
    seqs_arr = np.asarray(rows, np.int32).reshape(-1, L)
    pad = np.flatnonzero(seqs_arr.reshape(-1) == -1).astype(np.int32)
    return np.asarray(codes, np.int32), seqs_arr, pad, np.asarray(labels, np.float64)
