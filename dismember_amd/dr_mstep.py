"""Deep-Retrieval M-step: `CoordinateDescent.optimize`
(deep-retrieval/src/main/scala/com/mass/dr/optim/CoordinateDescent.scala:29-83) over the device beam search.

The expensive part — one beam search with `beam = numCandidatePath` per training sample (:143-147, :181-186) — is ONE
batched `dm_dr_beam_search` call per chunk; per-item aggregation is a sort + segmented sum over 64-bit path codes.  The greedy,
penalised path choice is inherently sequential over items (a shared path-size table), as in the reference, and stays on the host.
Orders the reference takes from hash maps are fixed here: ties between equal path scores resolve to the smaller path code
(ascending node tuple), items are visited in ascending id."""
import math

import numpy as np


def _codes(paths, K):
    c = np.zeros(paths.shape[:-1], np.int64)
    for d in range(paths.shape[-1]):
        c = c * K + paths[..., d]
    return c


def _decode(code, K, D):
    out = []
    for _ in range(D):
        out.append(int(code % K))
        code //= K
    return tuple(reversed(out))


def candidate_paths(engine, sequences, num_candidate_path, chunk=65536):
    """beamSearch(d.sequence, model, numCandidatePath) for every sample -> (path codes [N, C] int64 (-1 = none), probs [N, C])"""
    seqs = np.ascontiguousarray(sequences, np.int32)
    K = engine.dr_dims["K"]
    codes, probs = [], []
    for o in range(0, len(seqs), chunk):
        p, pr, cnt = engine.dr_beam_search(seqs[o:o + chunk], num_candidate_path)
        c = _codes(p.astype(np.int64), K)
        c[np.arange(p.shape[1])[None, :] >= cnt[:, None]] = -1
        codes.append(c)
        probs.append(pr)
    return np.concatenate(codes), np.concatenate(probs)


def batch_path_scores(engine, sequences, targets, num_candidate_path):
    """batchPathScore + aggregatePathScore (:117-163): {item: (codes [m] int64, scores [m] float64)}, scores descending."""
    codes, probs = candidate_paths(engine, sequences, num_candidate_path)
    tg = np.repeat(np.asarray(targets, np.int64), codes.shape[1])
    order_in = np.arange(codes.size)
    c, p = codes.ravel(), probs.ravel()
    keep = c >= 0
    tg, c, p, order_in = tg[keep], c[keep], p[keep], order_in[keep]
    o = np.lexsort((order_in, c, tg))                  # by item, then path, sums in sample order
    tg, c, p = tg[o], c[o], p[o]
    new = np.ones(len(c), bool)
    new[1:] = (tg[1:] != tg[:-1]) | (c[1:] != c[:-1])
    starts = np.flatnonzero(new)
    sums = np.add.reduceat(p, starts) if len(starts) else p[:0]     # pairwise inside numpy? no: reduceat is sequential
    gi, gc = tg[starts], c[starts]
    out = {}
    bounds = np.flatnonzero(np.r_[True, gi[1:] != gi[:-1], True])
    for a, b in zip(bounds[:-1], bounds[1:]):
        s, cc = sums[a:b], gc[a:b]
        k = np.argsort(-s, kind="stable")[:num_candidate_path]     # ties: ascending path code
        out[int(gi[a])] = (cc[k], s[k])
    return out


def streaming_path_scores(engine, sequences, targets, num_candidate_path, decay_factor=0.999, batch_size=8192):
    """streamingPathScore (:165-211): exponentially decayed running scores, merged sample by sample."""
    codes, probs = candidate_paths(engine, sequences, num_candidate_path)
    scores = {}
    for i, item in enumerate(np.asarray(targets).tolist()):
        m = codes[i] >= 0
        cand_c, cand_p = codes[i][m], probs[i][m]
        if item not in scores:
            scores[item] = (cand_c.copy(), cand_p.copy())
            continue
        oc, op = scores[item]
        min_score = op.min()
        union = np.union1d(oc, cand_c)                              # ascending path code
        so = dict(zip(oc.tolist(), op.tolist()))
        sc = dict(zip(cand_c.tolist(), cand_p.tolist()))
        new = np.array([decay_factor * so[u] + sc[u] if (u in so and u in sc)
                        else (decay_factor * min_score + sc[u] if u in sc else decay_factor * so[u]) for u in union.tolist()])
        k = np.argsort(-new, kind="stable")[:num_candidate_path]
        scores[item] = (union[k], new[k])
    return scores


def penalty_func(path_size, poly_order):
    f = lambda s: math.pow(s, poly_order) / poly_order
    return f(path_size + 1) - f(path_size)


def assign_paths(item_path_scores, item_occurrence, all_items, num_iteration, num_path_per_item, K, D, seed=0,
                 penalty_factor=3e-6, penalty_poly_order=4):
    """The coordinate-descent loop of `optimize` (:48-82) -> {item: [J path tuples]}."""
    rng = np.random.default_rng(seed)
    mapping, path_size = {}, {}
    for t in range(1, num_iteration + 1):
        for v in all_items:
            v = int(v)
            if v not in item_occurrence:
                mapping[v] = [int(c) for c in _codes(rng.integers(0, K, size=(num_path_per_item, D)), K)]   # generateRandomPath
                continue
            cc, ss = item_path_scores[v]
            selected, partial = [], 0.0
            nv = item_occurrence[v]
            for j in range(num_path_per_item - 1, -1, -1):
                if t > 1:
                    last = mapping[v][j]
                    path_size[last] = path_size[last] - 1
                best, best_score = None, None
                for code, prob in zip(cc.tolist(), ss.tolist()):
                    if code in selected:
                        continue
                    penalty = penalty_factor * penalty_func(path_size.get(code, 0), penalty_poly_order)
                    g = nv * (math.log1p(prob + partial) - math.log1p(partial)) - penalty
                    if best is None or g > best_score:
                        best, best_score = code, g
                if best is None:
                    raise ValueError("item %d has fewer candidate paths than paths per item" % v)
                path_size[best] = path_size.get(best, 0) + 1
                selected = [best] + selected
                partial = partial + best_score        # the reference accumulates the gain here (CoordinateDescent.scala:76)
            mapping[v] = selected
    return {v: [_decode(c, K, D) for c in codes] for v, codes in mapping.items()}


def optimize(engine, sequences, targets, all_items, num_candidate_path, num_path_per_item, num_iteration=3, train_mode="batch",
             decay_factor=0.999, batch_size=8192, seed=0, penalty_factor=3e-6, penalty_poly_order=4):
    """CoordinateDescent.optimize(model, trainMode) -> item -> paths (internal ids)."""
    K, D = engine.dr_dims["K"], engine.dr_dims["D"]
    if train_mode == "batch":
        sc = batch_path_scores(engine, sequences, targets, num_candidate_path)
    elif train_mode == "streaming":
        sc = streaming_path_scores(engine, sequences, targets, num_candidate_path, decay_factor, batch_size)
    else:
        raise ValueError(train_mode)
    tg, cnt = np.unique(np.asarray(targets), return_counts=True)          # computeItemOccurrence (:128-131)
    occ = dict(zip(tg.tolist(), cnt.tolist()))
    return assign_paths(sc, occ, sorted(int(i) for i in all_items), num_iteration, num_path_per_item, K, D, seed,
                        penalty_factor, penalty_poly_order)
