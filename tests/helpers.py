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


# ---------------------------------------------------------------------------------------------------------------------------
# "Every differing user is a near-tie at a cut", measured.  Two searches of the same user (device vs CPU oracle, or the two device
# arithmetics) that return different id lists agree on every tree index until the first prune (Recommender.scala:74-87: stable sort by
# -pred, take `beam`; TDM.scala:21 for the final top-k) at which their ORDERED kept lists differ.  At that cut both sides rank the same
# candidates by scores that differ only by rounding, so every pair of candidates the two sides order differently must be closer than
# the rounding allowance on BOTH sides.
def cut_disagreement(sa, sb, k):
    """Scores of the same candidates on two sides, cut = first k of the stable descending order.  None when the ordered kept lists
    coincide; else the largest |score gap| (on side a, on side b) over candidate pairs the sides order differently with at least one
    of the pair kept somewhere, the largest |score| among them, and how many candidates change place."""
    sa = np.asarray(sa, np.float64); sb = np.asarray(sb, np.float64)
    n = sa.size
    oa = np.argsort(-sa, kind="stable"); ob = np.argsort(-sb, kind="stable")
    if np.array_equal(oa[:k], ob[:k]):
        return None
    ra = np.empty(n, np.int64); rb = np.empty(n, np.int64)
    ra[oa] = np.arange(n); rb[ob] = np.arange(n)
    inv = np.flatnonzero((ra != rb) & ((ra < k) | (rb < k)))
    da = ra[inv][:, None] - ra[None, :]
    db = rb[inv][:, None] - rb[None, :]
    disc = (da * db) < 0                                   # ordered differently by the two sides
    ga = np.abs(sa[inv][:, None] - sa[None, :])[disc]
    gb = np.abs(sb[inv][:, None] - sb[None, :])[disc]
    mag = max(float(np.abs(sa[inv]).max()), float(np.abs(sb[inv]).max()))
    return dict(gap_a=float(ga.max()) if ga.size else 0.0, gap_b=float(gb.max()) if gb.size else 0.0, mag=mag, moved=int(inv.size))


def explain_divergence(otree, beam, topk, levels_a, levels_b, atol, rtol):
    """levels_x: [(candidate codes, scores)] per scored level of ONE user's search, as the traces give them (dm_tdm_beam_search_trace;
    oracle recommend(trace=True)).  Replays the reference's integer logic (otree.level_step / finalize = Recommender.scala:58-92,
    TDM.scala:21) on each side's own scores up to the first cut whose outcome differs.  Returns None when the sides never diverge, else
    dict(level = index of the scored level whose cut diverged, or 'final'; gap_a, gap_b, tol, score_diff, explained)."""
    start, level = (1 << (beam.bit_length() - 1)) - 1, beam.bit_length() - 1
    if not hasattr(otree, "_present"):
        otree._present = set(np.asarray(otree.codes).tolist())
    cand = np.array([c for c in range(start, 2 * start + 1) if c in otree._present], np.int32)      # Recommender.scala:53-56
    pa = np.zeros(cand.size, np.float32); pb = pa.copy()
    n_iter = otree.max_level - level + 1
    leaves_a, leaves_b = [], []

    def verdict(where, sa, sb, k):
        d = cut_disagreement(sa, sb, k)
        if d is None:
            return None
        tol = atol + rtol * d["mag"]
        diff = float(np.abs(np.asarray(sa, np.float64) - np.asarray(sb, np.float64)).max())
        return dict(level=where, gap_a=d["gap_a"], gap_b=d["gap_b"], tol=tol, score_diff=diff, moved=d["moved"],
                    explained=bool(d["gap_a"] <= 2 * tol and d["gap_b"] <= 2 * tol and diff <= tol))
    for it in range(n_iter):
        if cand.size > beam:
            v = verdict(it - 1, pa, pb, beam)
            if v is not None:
                return v
        la, lpa, cha = otree.level_step(beam, cand, pa)
        lb, lpb, chb = otree.level_step(beam, cand, pb)
        leaves_a.insert(0, (la, lpa)); leaves_b.insert(0, (lb, lpb))
        if not np.array_equal(cha, chb) or not np.array_equal(la, lb):
            # n <= beam never sorts, so a difference here without a cut verdict would be a logic bug, not a near-tie
            return dict(level=it - 1, gap_a=float("inf"), gap_b=float("inf"), tol=0.0, score_diff=float("inf"), moved=-1, explained=False)
        if cha.size == 0 or it >= len(levels_a) or it >= len(levels_b):
            break
        if not (np.array_equal(levels_a[it][0], cha) and np.array_equal(levels_b[it][0], cha)):
            return dict(level=it, gap_a=float("inf"), gap_b=float("inf"), tol=0.0, score_diff=float("inf"), moved=-1, explained=False)
        cand, pa, pb = cha, np.asarray(levels_a[it][1], np.float32), np.asarray(levels_b[it][1], np.float32)
    fa_c = np.concatenate([a for a, _ in leaves_a]) if leaves_a else np.zeros(0, np.int32)
    fa_p = np.concatenate([b for _, b in leaves_a]) if leaves_a else np.zeros(0, np.float32)
    fb_c = np.concatenate([a for a, _ in leaves_b]) if leaves_b else np.zeros(0, np.int32)
    fb_p = np.concatenate([b for _, b in leaves_b]) if leaves_b else np.zeros(0, np.float32)
    if not np.array_equal(fa_c, fb_c):
        return dict(level="final", gap_a=float("inf"), gap_b=float("inf"), tol=0.0, score_diff=float("inf"), moved=-1, explained=False)
    return verdict("final", fa_p, fb_p, topk)


def explain_users(otree, beam, topk, traces_a, traces_b, atol=1e-5, rtol=1e-4):
    """traces_x: per user a list [(codes, scores)] per scored level.  Summary over the users that diverge:
    dict(differing_users, explained_by_near_tie, max_cut_gap, max_cut_gap_over_tol, levels histogram, worst)."""
    out = dict(differing_users=0, explained_by_near_tie=0, max_cut_gap=0.0, max_cut_gap_over_tol=0.0, by_level={}, unexplained=[])
    for u, (la, lb) in enumerate(zip(traces_a, traces_b)):
        v = explain_divergence(otree, beam, topk, la, lb, atol, rtol)
        if v is None:
            continue
        out["differing_users"] += 1
        out["by_level"][str(v["level"])] = out["by_level"].get(str(v["level"]), 0) + 1
        if v["explained"]:
            out["explained_by_near_tie"] += 1
            g = max(v["gap_a"], v["gap_b"])
            out["max_cut_gap"] = max(out["max_cut_gap"], g)
            out["max_cut_gap_over_tol"] = max(out["max_cut_gap_over_tol"], g / (2 * v["tol"]))
        else:
            out["unexplained"].append((u, v))
    return out


def trace_levels(tc, ts, tn, u):
    """One user's [(codes, scores)] from the arrays dm_tdm_beam_search_trace fills."""
    return [(tc[u, it, :int(tn[u, it])].copy(), ts[u, it, :int(tn[u, it])].copy()) for it in range(tn.shape[1]) if int(tn[u, it]) > 0]
