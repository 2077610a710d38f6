"""OTM tree construction (row A9, the OTM twin): oracle known answers (CPU) and the device path against it (GPU)."""
import numpy as np
import pytest

from oracle import otm_tree_oracle as oo


def test_helpers_known_answers():
    assert oo.get_children_at_level(0, 0, 2) == [3, 4, 5, 6] and oo.get_children_at_level(2, 1, 2) == [5, 6]
    assert oo.get_ancestor_at_level(12, 1) == 2 and oo.get_ancestor_at_level(12, 3) == 12
    assert oo.sort_node_weights([0.1, 0.7, 0.7, -1.0], [3, 4, 5, 6]) == [(4, 0.7), (5, 0.7), (3, 0.1), (6, -1.0)]   # stable


def test_rebalance_known_answer():
    # 4 items all prefer child 3 (capacity 2): the two that were already under 3 stay, the others go to their second choice
    children = [3, 4]
    cand = {1: [(3, 0.9), (4, 0.1)], 2: [(3, 0.8), (4, 0.2)], 3: [(3, 0.7), (4, 0.3)], 4: [(3, 0.6), (4, 0.4)]}
    node_items = {3: [(i, cand[i][0][1], 1) for i in (1, 2, 3, 4)]}
    old = {1: 4, 2: 3, 3: 4, 4: 3}
    res = oo.re_balance(node_items, old, children, 2, cand)
    assert sorted(i for i, _, _ in res[3]) == [2, 4] and sorted(i for i, _, _ in res[4]) == [1, 3]
    # without "stayed" preferences the two best weights stay
    res = oo.re_balance(node_items, {i: 9 for i in cand}, children, 2, cand)
    assert sorted(i for i, _, _ in res[3]) == [1, 2]


@pytest.mark.gpu
def test_otm_tree_construction_vs_oracle(fixture_w64, fixture_otm_mapping, oracle):
    from dismember_amd import Engine
    from dismember_amd.otm_tree import TreeConstruction
    E, L = 16, 10
    rng = np.random.default_rng(14)
    omap = fixture_otm_mapping.astype(np.int64)
    sel = np.sort(rng.choice(len(omap), 200, replace=False))             # 200 items -> leaf level 8
    # re-map the chosen items onto the leaves of an 8-level tree (the bundled mapping is 12 levels deep)
    leaf_level = 8
    leaves = (1 << leaf_level) - 1 + rng.permutation(1 << leaf_level)[:200]
    item_leaf = {int(omap[s, 0]): int(l) for s, l in zip(sel, leaves)}
    items = sorted(item_leaf)
    seqs = {}
    for it in items[:170]:                                               # 30 items never appear as a target
        n = int(rng.integers(1, 5))
        r = rng.choice(leaves, size=(n, L)).astype(np.int32)
        r[rng.random((n, L)) < 0.2] = -1
        seqs[it] = r.reshape(-1)
    eng = Engine(0)
    eng.load_weights_din(fixture_w64, E, 8191)
    din = oracle.Din(fixture_w64, E, L, 8191)
    tc = TreeConstruction(eng, item_leaf, seqs, gap=2, seq_len=L)
    assert tc.leaf_level == leaf_level
    # one gap step: weights vs the oracle (fp64 model: 1e-9), assignment bit-exact on the same weights
    old_level, level = 2, 4
    node_of = TreeConstruction.ancestor_at_level(tc.item_leaf, old_level)
    w = tc.child_weights(node_of, old_level, level)
    for k in rng.choice(len(items), 25, replace=False):
        ch = oo.get_children_at_level(int(node_of[k]), old_level, level)
        want = [oo.aggregate_weights(din, seqs.get(items[k]), L, int(node_of[k]), c) for c in ch]
        np.testing.assert_allclose(w[k], want, rtol=1e-9, atol=1e-9)
    parts = [tc.weights_range(node_of, old_level, level, a, b) for a, b in ((0, 70), (70, 71), (71, 200))]
    assert np.array_equal(np.concatenate(parts), w)                     # item shards reproduce the matrix bit for bit
    for node in np.unique(node_of)[:4]:
        grp = np.flatnonzero(node_of == node)
        old_node = TreeConstruction.ancestor_at_level(tc.item_leaf[grp], level)
        got = tc.rebalance(w[grp], old_node, int(node), old_level, level, 1 << (leaf_level - level))
        want = oo.get_children_projection(din, seqs, L, item_leaf, leaf_level, old_level, level, int(node), [items[i] for i in grp], weights=w[grp])
        assert {items[i]: int(g) for i, g in zip(grp, got) if g >= 0} == want
    # the whole run: a bijection onto distinct leaves (TreeConstructionSpec.scala:38-48), equal to the oracle-driven run
    proj = tc.run()
    codes = np.array(list(proj.values()))
    assert set(proj) == set(items) and codes.min() >= 255 and codes.max() <= 510 and len(set(codes.tolist())) == 200
    ref = oo.run(din, item_leaf, seqs, L, 2)
    same = sum(int(proj[i] == ref[i]) for i in items)
    assert same >= 0.97 * len(items), same
