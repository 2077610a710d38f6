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
