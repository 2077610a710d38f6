"""Synthetic catalogue / tree / user generators for the configurations in BASELINE.json
(SURVEY.md §8d): numpy only, seeded, nothing from the reference."""
import numpy as np

SEED = 20250523


def make_tree(n_items, depth, rng):
    """Complete heap to `depth`; the last 2^depth - n_items leaf codes are absent and their
    childless ancestors pruned.  Item id = 1 + permutation index (0 is padding)."""
    first = (1 << depth) - 1
    assert 0 < n_items <= (1 << depth)
    leaf_codes = np.arange(first, first + n_items, dtype=np.int64)
    levels = [leaf_codes]
    c = leaf_codes
    for _ in range(depth):
        c = np.unique((c - 1) >> 1)
        levels.append(c)
    anc = np.concatenate(levels[1:][::-1])
    leaf_ids = (rng.permutation(n_items) + 1).astype(np.int32)
    non_leaf_offset = int(n_items) + 1
    codes = np.concatenate([anc, leaf_codes]).astype(np.int32)
    ids = np.concatenate([(anc + non_leaf_offset).astype(np.int32), leaf_ids])
    is_leaf = np.concatenate([np.zeros(anc.size, np.uint8), np.ones(n_items, np.uint8)])
    return dict(codes=codes, ids=ids, is_leaf=is_leaf, leaf_ids=leaf_ids, leaf_codes=leaf_codes.astype(np.int32),
                max_level=depth)


def make_din_weights(E, num_index, rng, dtype=np.float32):
    """Reference init: every matrix N(0, 0.05), biases 0 (Linear.scala:12-13, EmbeddingShare.scala:19-22)."""
    n = num_index * E + 3 * E * E + 2 * E + 1
    w = np.empty(n, dtype)
    chunk = 1 << 24
    for o in range(0, num_index * E + 3 * E * E, chunk):
        m = min(chunk, num_index * E + 3 * E * E - o)
        w[o:o + m] = rng.standard_normal(m, dtype=np.float32) * 0.05
    off = num_index * E + 3 * E * E
    w[off:off + E] = 0                        # l1 bias
    w[off + E:off + 2 * E] = rng.standard_normal(E, dtype=np.float32) * 0.05   # l2 weight
    w[off + 2 * E] = 0                        # l2 bias
    return w


def make_users(leaf_ids, n_users, L, rng, zipf_s=1.0, pad_p=0.15):
    """Histories of item ids: Zipf(zipf_s) over the items, Binomial(L, pad_p) leading pads."""
    n = leaf_ids.size
    pmf = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(pmf)
    cdf /= cdf[-1]
    idx = np.searchsorted(cdf, rng.random((n_users, L)))
    seq = leaf_ids[np.minimum(idx, n - 1)].astype(np.int32)
    npad = rng.binomial(L, pad_p, n_users)
    seq[np.arange(L)[None, :] < npad[:, None]] = 0
    return seq


def make_dr_model(num_item, K, D, L, E, rng, scale=0.3, dtype=np.float64):
    """Deep-Retrieval weights with the reference's shapes (deep-retrieval/.../model/LayerModel.scala:22-39,
    RerankModel.scala:13-35): layer embedding rows = num_item + K(D-1), layer d Linear((L+d)E -> K),
    rerank Embedding(num_item, E) + Linear(L*E -> E) + softmaxWeights/softmaxBiases.  `scale` is the std of every
    matrix (the reference initialises with 0.05; a wider spread gives a peaked, "trained-like" path distribution)."""
    n = lambda *s: (rng.standard_normal(s) * scale).astype(dtype)
    return dict(
        layer_emb=n(num_item + K * (D - 1), E),
        layer_w=[n(K, (L + d) * E) for d in range(D)],
        layer_b=[(rng.standard_normal(K) * 0.1).astype(dtype) for _ in range(D)],
        rerank_emb=n(num_item, E), rerank_w=n(E, L * E), rerank_b=(rng.standard_normal(E) * 0.1).astype(dtype),
        softmax_w=n(num_item, E), softmax_b=(rng.standard_normal(num_item) * 0.1).astype(dtype))


def make_dr_paths(num_item, K, D, J, rng):
    """J random paths per item (MappingOp.initItemPathMapping, deep-retrieval/.../model/MappingOp.scala:30-43)."""
    return rng.integers(0, K, size=(num_item, J, D), dtype=np.int32)


def dr_path_items(item_paths, collapse=False):
    """item -> paths [num_item, J, D]  =>  path -> items CSR (distinct paths sorted lexicographically).

    collapse=False keeps every item of a path in ascending id order (the intended inverse map);
    collapse=True keeps one item per path, the LAST in ascending id order — the shape MappingOp.pathToItems
    (MappingOp.scala:23-28) really produces, where a Map-typed flatMap overwrites earlier items of the same
    path (which item survives there depends on HashMap iteration order, so only the shape is reproducible)."""
    n, J, D = item_paths.shape
    flat = item_paths.reshape(n * J, D).astype(np.int64)
    item = np.repeat(np.arange(n, dtype=np.int32), J)
    order = np.lexsort((item,) + tuple(flat.T[::-1]))
    flat, item = flat[order], item[order]
    new = np.ones(len(flat), bool)
    new[1:] = (flat[1:] != flat[:-1]).any(axis=1)
    # an item listing the same path twice contributes once (Map key semantics)
    dup = np.zeros(len(flat), bool)
    dup[1:] = (~new[1:]) & (item[1:] == item[:-1])
    flat, item, new = flat[~dup], item[~dup], new[~dup]
    starts = np.flatnonzero(new)
    off = np.concatenate([starts, [len(flat)]]).astype(np.int64)
    paths = flat[starts].astype(np.int32)
    if collapse:
        item = item[off[1:] - 1]
        off = np.arange(len(paths) + 1, dtype=np.int64)
    return paths, off, item.astype(np.int32)


def dr_path_items_fast(item_paths, K):
    """dr_path_items(collapse=False) for large catalogues: path codes + one stable argsort instead of a lexsort."""
    n, J, D = item_paths.shape
    code = np.zeros(n * J, np.int64)
    for d in range(D):
        code = code * K + item_paths[:, :, d].reshape(-1)
    item = np.repeat(np.arange(n, dtype=np.int32), J)
    order = np.argsort(code, kind="stable")
    code, item = code[order], item[order]
    new = np.ones(len(code), bool)
    new[1:] = code[1:] != code[:-1]
    dup = np.zeros(len(code), bool)
    dup[1:] = (~new[1:]) & (item[1:] == item[:-1])
    code, item, new = code[~dup], item[~dup], new[~dup]
    starts = np.flatnonzero(new)
    off = np.concatenate([starts, [len(code)]]).astype(np.int64)
    c = code[starts]
    paths = np.empty((len(c), D), np.int32)
    for d in range(D - 1, -1, -1):
        paths[:, d] = c % K
        c = c // K
    return paths, off, item


def make_tree_consistent_interactions(leaf_ids, n_users, L, rng, spread=64.0, pad_p=0.15):
    """(histories, targets) whose structure FOLLOWS the tree: a user is an anchor position in the leaf order, its history items sit a
    Laplace(spread) number of leaves away from the anchor and its target a Laplace(spread / 4) number away — so an ancestor of the
    target is, level by level, the subtree most of the history lies in (what TDM's training makes the scorer learn,
    T/dataset/TDMTrainSet + NegativeSampler: positives = the target's ancestors).  leaf_ids is in leaf-code order (make_tree)."""
    n = leaf_ids.size
    anchor = rng.integers(0, n, n_users)
    off = np.rint(rng.laplace(0.0, spread, (n_users, L))).astype(np.int64)
    seq = leaf_ids[np.clip(anchor[:, None] + off, 0, n - 1)].astype(np.int32)
    npad = rng.binomial(L, pad_p, n_users)
    seq[np.arange(L)[None, :] < npad[:, None]] = 0
    toff = np.rint(rng.laplace(0.0, spread / 4.0, n_users)).astype(np.int64)
    tgt = leaf_ids[np.clip(anchor + toff, 0, n - 1)].astype(np.int32)
    return seq, tgt
