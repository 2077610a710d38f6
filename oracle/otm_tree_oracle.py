"""TEST INFRASTRUCTURE — restatement of OTM's tree construction (the OTM twin), pure Python
loops over the C fp64 DIN forward.  O/ = /root/reference/otm/src/main/scala/com/mass/otm/.

  aggregate_weights / sort_node_weights   O/tree/TreeConstruction.scala:180-212 (+ buildFeatures :214-232)
  re_balance / get_max_node               O/tree/TreeConstruction.scala:287-352
  get_children_projection / run           O/tree/TreeConstruction.scala:44-141
Iteration orders the reference takes from hash maps (`itemIdMapping.keys`, `groupMap`) are replaced by ascending item id.
Parity status: unpinned at the JVM boundary (otm/src/test/scala/TreeConstructionSpec.scala:38-48 checks the leaf range
only); this file is pinned on those invariants and on hand-checkable cases in tests/test_otm_tree.py.
"""
import functools
import struct

import numpy as np


def java_double_compare(x, y):
    """java.lang.Double.compare (what Ordering[Double] resolves to): numeric order, then doubleToLongBits — -0.0 < 0.0, one
    canonical NaN above everything."""
    if x < y:
        return -1
    if x > y:
        return 1
    a = struct.unpack("<q", struct.pack("<d", float("nan") if x != x else float(x)))[0] if x == x else 0x7ff8000000000000
    b = struct.unpack("<q", struct.pack("<d", float("nan") if y != y else float(y)))[0] if y == y else 0x7ff8000000000000
    return 0 if a == b else (-1 if a < b else 1)


def get_ancestor_at_level(node, level):
    lim = (1 << (level + 1)) - 1
    while node >= lim:
        node = (node - 1) >> 1
    return node


def get_children_at_level(ancestor, old_level, level):
    nodes = [ancestor]
    for _ in range(old_level, level):
        nodes = [c for n in nodes for c in (2 * n + 1, 2 * n + 2)]
    return nodes


def aggregate_weights(din, item_seq, L, current_node, child_node, use_mask=True):
    """item_seq: flat [rows * L] node ids of one item (None: the item never appeared as a target)."""
    if item_seq is None:
        return -1e6
    seq = np.asarray(item_seq, np.int32).reshape(-1, L)
    weights = 0.0
    node = child_node
    while node > current_node:
        pad = np.flatnonzero(seq.reshape(-1) == -1).astype(np.int32) if use_mask else None
        out = din.forward(np.full(len(seq), node, np.int32), seq, pad)
        score = 0.0
        for v in out:                   # Tensor.sum, sequential
            score += float(v)
        weights += score
        node = (node - 1) >> 1
    return weights


def sort_node_weights(weights_row, children):
    order = sorted(range(len(children)), key=functools.cmp_to_key(lambda a, b: java_double_compare(weights_row[b], weights_row[a])))
    return [(children[i], weights_row[i]) for i in order]


def re_balance(node_items, old_item_node, children, max_assign, cand):
    """node_items: {child: [(item, weight, next_idx)]}; cand: {item: [(child, weight)] sorted}."""
    res = {k: list(v) for k, v in node_items.items()}
    processed = set()
    while True:
        best = None
        for n in children:                                    # getMaxNode: maxBy over (count, node) pairs, first maximum
            v = (len(res[n]), n) if (n not in processed and n in res) else (-1, 0)
            if best is None or v[0] > best[0]:
                best = v
        if best[0] <= max_assign:
            return res
        node = best[1]
        processed.add(node)
        srt = sorted(res[node], key=functools.cmp_to_key(
            lambda a, b: (int(old_item_node[a[0]] != node) - int(old_item_node[b[0]] != node)) or java_double_compare(b[1], a[1])))
        res[node] = srt[:max_assign]
        for item, _, nxt in srt[max_assign:]:
            cw = cand[item]
            idx = nxt
            while idx < len(cw):
                n2, w2 = cw[idx]
                if n2 not in processed:
                    res.setdefault(n2, []).append((item, w2, idx + 1))
                    break
                idx += 1


def get_children_projection(din, item_seqs, L, item_leaf, leaf_level, old_level, level, node, items, use_mask=True, weights=None):
    max_assign = 1 << (leaf_level - level)
    children = get_children_at_level(node, old_level, level)
    cand = {}
    for k, item in enumerate(items):
        row = [aggregate_weights(din, item_seqs.get(item), L, node, c, use_mask) for c in children] if weights is None else list(weights[k])
        cand[item] = sort_node_weights(row, children)
    node_items = {}
    for item in items:
        c, w = cand[item][0]
        node_items.setdefault(c, []).append((item, w, 1))
    old_item_node = {item: get_ancestor_at_level(item_leaf[item], level) for item in items}
    out = {}
    for n, lst in re_balance(node_items, old_item_node, children, max_assign, cand).items():
        assert len(lst) <= max_assign
        for item, _, _ in lst:
            out[item] = n
    return out


def run(din, item_leaf, item_seqs, L, gap, use_mask=True):
    """TreeConstruction.run: item -> new leaf node."""
    leaf_level = int(np.ceil(np.log(len(item_leaf)) / np.log(2)))
    proj = {item: 0 for item in sorted(item_leaf)}
    for old_level in range(0, leaf_level, gap):
        level = min(leaf_level, old_level + gap)
        groups = {}
        for item in sorted(proj):
            groups.setdefault(proj[item], []).append(item)
        for node in sorted(groups):
            proj.update(get_children_projection(din, item_seqs, L, item_leaf, leaf_level, old_level, level, node, groups[node], use_mask))
    return proj
