"""Shared synthetic-data builders for the parity tests (numpy only)."""
import numpy as np

CANONICAL_TDM_QUERY = [0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882]  # examples/.../tdm/package.scala:115


def din_param_count(E, num_index):
    return num_index * E + E * E + 2 * E * E + E + E + 1


def random_din_weights(rng, E, num_index, dtype=np.float32, std=0.05, bias_std=0.05):
    """Compact A0 vector with the reference's init scale (N(0, 0.05)); biases non-zero on purpose."""
    w = rng.normal(0.0, std, din_param_count(E, num_index)).astype(dtype)
    off = num_index * E + 3 * E * E
    w[off:off + 2 * E + 1] = rng.normal(0.0, bias_std, 2 * E + 1).astype(dtype)
    return w


def synthetic_tree(rng, depth, n_items):
    """Complete heap to `depth` with only the first n_items leaf codes present (ancestors of
    absent leaves pruned), item ids = 1 + a random permutation.  Returns dict like the fixture."""
    first = (1 << depth) - 1
    assert 0 < n_items <= (1 << depth)
    leaf_codes = np.arange(first, first + n_items, dtype=np.int64)
    present = set(leaf_codes.tolist())
    c = leaf_codes
    for _ in range(depth):
        c = np.unique((c - 1) >> 1)
        present.update(c.tolist())
    codes = np.array(sorted(present), dtype=np.int32)
    perm = rng.permutation(n_items).astype(np.int32) + 1
    leaf_ids = perm
    non_leaf_offset = int(leaf_ids.max()) + 1
    ids = codes + non_leaf_offset            # ancestors: id = code + offset (TreeBuilder convention)
    is_leaf = (codes >= first).astype(np.uint8)
    lut = {int(cc): int(i) for cc, i in zip(leaf_codes, leaf_ids)}
    ids = np.array([lut.get(int(cc), int(cc) + non_leaf_offset) for cc in codes], dtype=np.int32)
    return dict(codes=codes, ids=ids, is_leaf=is_leaf, leaf_ids=leaf_ids.astype(np.int32),
                leaf_codes=leaf_codes.astype(np.int32), max_level=np.int32(depth))


def random_histories(rng, leaf_ids, U, L, pad_prob=0.15, unknown_prob=0.0):
    """Item-id histories with prefix padding (0) like TreeInit produces (TreeInit.scala:253,285)."""
    seq = rng.choice(leaf_ids, size=(U, L)).astype(np.int32)
    npad = rng.binomial(L, pad_prob, size=U)
    for u in range(U):
        seq[u, :npad[u]] = 0
    if unknown_prob > 0:
        m = rng.random((U, L)) < unknown_prob
        seq[m] = 2 ** 30   # id far beyond nonLeafOffset + maxCode -> padded + masked by idToCode
    return seq


def numpy_din_forward(w, E, L, num_index, codes, seqs, pad_flat):
    """Independent float64 numpy statement of the DIN math (used to cross-check the C oracle)."""
    w = w.astype(np.float64)
    emb = w[:num_index * E].reshape(num_index, E)
    o = num_index * E
    att_w = w[o:o + E * E].reshape(E, E); o += E * E
    l1_w = w[o:o + 2 * E * E].reshape(E, 2 * E); o += 2 * E * E
    l1_b = w[o:o + E]; o += E
    l2_w = w[o:o + E]; o += E
    l2_b = w[o]
    codes = np.asarray(codes); seqs = np.asarray(seqs).reshape(len(codes), L)
    q = np.where(codes[:, None] >= 0, emb[np.maximum(codes, 0)], 0.0)
    k = np.where(seqs[..., None] >= 0, emb[np.maximum(seqs, 0)], 0.0)
    s = np.einsum("be,ble->bl", q, k) / np.sqrt(E)
    flat = s.reshape(-1)
    flat[np.asarray(pad_flat, dtype=np.int64)] = -np.finfo(np.float32).max
    s = flat.reshape(len(codes), L)
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    c = np.einsum("bl,ble->be", p, k)
    a = c @ att_w.T
    h = np.concatenate([q, a], -1) @ l1_w.T + l1_b
    h = np.maximum(h, 0)
    return h @ l2_w + l2_b
