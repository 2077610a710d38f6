"""Device-side level-wise negative sampling (row A10): dm_tdm_make_train_batch / dm_tdm_sample_train_batch_dev against
  * the CPU oracle's restatement of NegativeSampler.sample (tdm/.../utils/NegativeSampler.scala:76-158) on the same
    counter-based stream — BIT-EXACT rows, uniform and sample_with_probability modes, ragged targets, sparse levels;
  * the distributions the reference samples from (its own RNG is unseeded, so that is all the parity there is): a
    chi-square test per level against the uniform / node-probability law.
"""
import numpy as np
import pytest

from helpers import random_histories, synthetic_tree

pytestmark = pytest.mark.gpu
NEG = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17, 19, 22, 25, 30, 76, 200], np.int32)   # model.layer_negative_counts


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("with_prob", [False, True])
def test_device_sampler_equals_oracle_restatement(engine_fixture, oracle, oracle_tree, fixture_tree, with_prob):
    rng = np.random.default_rng(41)
    T, L = 57, 10
    t = fixture_tree
    seqs = random_histories(rng, t["leaf_ids"], T, L, unknown_prob=0.05)
    tgt = rng.choice(t["leaf_ids"], T).astype(np.int32)
    tgt[3] = 0                    # padding target: no rows (pathNodes is empty)
    tgt[9] = 2 ** 30              # unknown id: no rows
    if with_prob:
        engine_fixture.set_node_probs(t["codes"], t["probs"])
    for start, seed, tol in [(1, 7, 20), (4, 123456789, 0), (12, 5, 3)]:
        got = engine_fixture.make_train_batch(seqs, tgt, NEG, start_level=start, seed=seed, with_prob=with_prob, tolerance=tol)
        want = oracle.tdm_sample_batch(oracle_tree, seqs, tgt, NEG, start_level=start, seed=seed, with_prob=with_prob, tolerance=tol,
                                       node_codes=t["codes"], node_probs=t["probs"])
        per = sum(1 + int(NEG[l]) for l in range(start, 13))
        assert got[0].size == want[0].size == (T - 2) * per
        assert _same(got, want), (start, seed)
    # use_mask = False: no mask bits, same codes
    a = engine_fixture.make_train_batch(seqs, tgt, NEG, seed=7, use_mask=False, with_prob=with_prob)
    b = oracle.tdm_sample_batch(oracle_tree, seqs, tgt, NEG, seed=7, use_mask=False, with_prob=with_prob, node_codes=t["codes"],
                                node_probs=t["probs"])
    assert _same(a, b) and (a[2] == 0).all()


def test_device_sampler_sparse_and_ragged_tree(oracle):
    """A tree whose last level is sparse (fewer existing nodes than negatives + 1 on some levels -> the bounded retry gives up,
    as in the oracle) and whose leaves sit on two levels (targets of different depth -> different row counts)."""
    from dismember_amd import Engine
    rng = np.random.default_rng(6)
    t = synthetic_tree(rng, 9, 37)                 # 37 of 512 leaves: level 9 holds 37 nodes, level 8 19, ...
    codes, ids, is_leaf = t["codes"].copy(), t["ids"].copy(), t["is_leaf"].copy()
    # turn the level-8 ancestor of the last leaf into a leaf of its own (a second, shallower leaf level)
    last = int(t["leaf_codes"][-1])
    if last % 2 == 1:                               # an only child: drop it and make the parent a leaf item
        par = (last - 1) >> 1
        keep = codes != last
        codes, ids, is_leaf = codes[keep], ids[keep], is_leaf[keep]
        is_leaf[codes == par] = 1
        leaf_ids = t["leaf_ids"].copy(); leaf_codes = t["leaf_codes"].copy()
        leaf_codes[-1] = par
        ids[codes == par] = leaf_ids[-1]
    else:
        leaf_ids, leaf_codes = t["leaf_ids"], t["leaf_codes"]
    eng = Engine(0)
    eng.load_tree(codes, ids, is_leaf, 9)
    eng.load_id_maps(leaf_ids, leaf_codes)
    otree = oracle.TdmTree(codes, ids, is_leaf, leaf_ids, leaf_codes, 9)
    from dismember_amd import DismemberError
    seqs = random_histories(rng, leaf_ids, 40, 7)
    tgt = rng.choice(leaf_ids, 40).astype(np.int32)
    tgt[0] = leaf_ids[-1]
    # level 8 holds 19 nodes: 30 distinct negatives besides the positive cannot exist (the reference's rejection loop would
    # never return); the library reports it instead of spinning
    with pytest.raises(DismemberError) as e:
        eng.make_train_batch(seqs, tgt, np.array([0, 0, 0, 0, 1, 2, 4, 7, 30, 60], np.int32), start_level=4, seed=1)
    assert e.value.code == -1 and "level 8 has only 19 nodes" in str(e.value)
    neg = np.array([0, 0, 0, 0, 1, 2, 4, 8, 17, 34], np.int32)       # nearly every node of the sparse levels (4: 1 of 2 ... 8: 17 of 19, 9: 34 of 36)
    got = eng.make_train_batch(seqs, tgt, neg, start_level=4, seed=99)
    want = oracle.tdm_sample_batch(otree, seqs, tgt, neg, start_level=4, seed=99)
    assert got[0].size > 0 and _same(got, want)
    probs = rng.random(codes.size).astype(np.float32) + 0.01
    eng.set_node_probs(codes, probs)
    got = eng.make_train_batch(seqs, tgt, neg, start_level=4, seed=5, with_prob=True, tolerance=4)
    want = oracle.tdm_sample_batch(otree, seqs, tgt, neg, start_level=4, seed=5, with_prob=True, tolerance=4, node_codes=codes, node_probs=probs)
    assert _same(got, want)
    eng.close()


def _chi2_ok(counts, expect, slack=6.0):
    """Pearson statistic against its mean (dof) with a generous bound: mean + slack * sqrt(2 dof)."""
    m = expect > 0
    stat = float((((counts[m] - expect[m]) ** 2) / expect[m]).sum())
    dof = int(m.sum()) - 1
    return stat <= dof + slack * np.sqrt(2.0 * dof), (stat, dof)


def test_sampling_distributions_per_level(engine_fixture, fixture_tree):
    """Uniform mode: every existing node of a level except the positive is equally likely.  Probability mode: first draws
    follow Node.probality (one negative per level, a huge tolerance: no uniform fill, no without-replacement distortion)."""
    t = fixture_tree
    rng = np.random.default_rng(17)
    T, L = 6000, 10
    seqs = np.zeros((T, L), np.int32)
    tgt = np.full(T, t["leaf_ids"][0], np.int32)                # one fixed target: the positive of every level is fixed too
    code = int(t["leaf_codes"][0])
    path = {}
    c = code
    while c > 0:
        path[int(np.floor(np.log2(c + 1)))] = c
        c = (c - 1) >> 1
    one = np.array([0] + [1] * 12, np.int32)
    present = np.zeros(8191, bool); present[t["codes"]] = True
    prob_of = np.zeros(8191, np.float64); prob_of[t["codes"]] = t["probs"]
    engine_fixture.set_node_probs(t["codes"], t["probs"])
    for with_prob in (False, True):
        codes, _, _, y = engine_fixture.make_train_batch(seqs, tgt, one, start_level=1, seed=int(rng.integers(1 << 40)), with_prob=with_prob,
                                                          tolerance=1000)
        negs = codes[y == 0]
        for level in (3, 6, 9, 12):
            lo, hi = 2 ** level - 1, 2 ** (level + 1) - 1
            sel = negs[(negs >= lo) & (negs < hi)]
            assert sel.size == T
            counts = np.bincount(sel - lo, minlength=hi - lo).astype(np.float64)
            ok_nodes = present[lo:hi].copy()
            ok_nodes[path[level] - lo] = False                   # the positive is never a negative here
            assert counts[~ok_nodes].sum() == 0
            if with_prob:
                w = prob_of[lo:hi] * ok_nodes                    # a rejected draw (the positive) is simply redrawn: renormalise
                expect = T * w / w.sum()
            else:
                expect = T * ok_nodes / ok_nodes.sum()
            # greedy bins of expected count >= 10 (in order of expectation), so the statistic is meaningful on wide levels
            order = np.argsort(expect, kind="stable")
            e_m, c_m, ea, ca = [], [], 0.0, 0.0
            for i in order:
                ea += expect[i]; ca += counts[i]
                if ea >= 10.0:
                    e_m.append(ea); c_m.append(ca); ea = ca = 0.0
            if ea > 0:
                e_m[-1] += ea; c_m[-1] += ca
            assert len(e_m) >= 5
            ok, info = _chi2_ok(np.array(c_m), np.array(e_m))
            assert ok, (with_prob, level, info)


def test_train_step_on_device_rows_equals_host_path(fixture_tree, fixture_w32):
    """convertBatch + trainBatch with the rows resident in HBM == the same rows going through the host entry points."""
    from dismember_amd import Engine
    t = fixture_tree
    rng = np.random.default_rng(23)
    T, L = 64, 10
    seqs = random_histories(rng, t["leaf_ids"], T, L)
    tgt = rng.choice(t["leaf_ids"], T).astype(np.int32)
    out = []
    for path in ("device", "host"):
        eng = Engine(0)
        eng.load_tree(t["codes"], t["ids"], t["is_leaf"], 12); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
        eng.load_weights_din(fixture_w32, 16, 8191)
        eng.train_init(lr=1e-3)
        if path == "device":
            loss = eng.train_step_sampled(seqs, tgt, NEG, start_level=1, seed=77)
        else:
            c, s, m, y = eng.make_train_batch(seqs, tgt, NEG, start_level=1, seed=77)
            loss = eng.train_forward_backward(c, s, eng.rowmask_to_flat(m, L), y)
        g = eng.train_download("grad")
        out.append((loss, g))
        eng.close()
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-6)
    # float atomics: the accumulation order differs from launch to launch, the values do not
    assert np.allclose(out[0][1], out[1][1], rtol=1e-4, atol=1e-7)
