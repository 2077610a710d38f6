"""Deep-Retrieval M-step: known answers for the oracle restatement, host mirror == restatement when
both see the same candidate scores (CPU), device path scores vs the fp64 oracle's (GPU)."""
import math

import numpy as np
import pytest

from dismember_amd import dr_mstep, synth
from oracle import dr_mstep_oracle as mo


def test_penalty_and_greedy_choice_known_answers():
    assert mo.penalty_func(0, 4) == 0.25 and mo.penalty_func(1, 4) == (16 - 1) / 4.0
    assert dr_mstep.penalty_func(2, 4) == mo.penalty_func(2, 4) == (81 - 16) / 4.0
    # one item, 3 candidate paths, J = 2, no penalty pressure: the two best paths, best first in the list
    ips = {5: [((0, 1), 0.5), ((2, 2), 0.3), ((1, 0), 0.1)]}
    m = mo.optimize(ips, {5: 4}, [5], 1, 2, lambda v: [], penalty_factor=0.0)
    g0 = 4 * math.log1p(0.5)
    assert m[5] == [(2, 2), (0, 1)]          # j = 1 picks (0,1) first; j = 0 then prepends its pick
    # a huge penalty on an already crowded path flips the choice
    ips = {1: [((0, 0), 0.9)], 2: [((0, 0), 0.9), ((1, 1), 0.8)]}
    m = mo.optimize(ips, {1: 1, 2: 1}, [1, 2], 1, 1, lambda v: [], penalty_factor=0.5, penalty_poly_order=1)
    assert m[1] == [(0, 0)] and m[2] == [(0, 0)]          # order-1 penalty is the same for every size: no flip
    m = mo.optimize(ips, {1: 1, 2: 1}, [1, 2], 1, 1, lambda v: [], penalty_factor=0.05, penalty_poly_order=4)
    assert m[2] == [(1, 1)]                                # size 1 -> penalty 0.05 * 15/4 outweighs 0.9 vs 0.8
    assert g0 > 0


def test_batch_and_streaming_scores_known_answers():
    table = {(1,): [((0, 0), 0.6), ((0, 1), 0.4)], (2,): [((0, 1), 0.7), ((1, 1), 0.3)]}
    bs = lambda seq, beam: table[tuple(seq)][:beam]
    samples = [([1], 9), ([2], 9), ([1], 4)]
    b = mo.batch_path_score(samples, bs, 2)
    assert b[9] == [((0, 1), 0.4 + 0.7), ((0, 0), 0.6)] and b[4] == [((0, 0), 0.6), ((0, 1), 0.4)]
    s = mo.streaming_path_score(samples, bs, 2, 0.5, 10)
    # item 9: first sample seeds; second: (0,1) in both -> 0.5*0.4+0.7, (0,0) only old -> 0.5*0.6, (1,1) only new -> 0.5*min(0.4)+0.3
    assert s[9] == [((0, 1), 0.5 * 0.4 + 0.7), ((1, 1), 0.5 * 0.4 + 0.3)]


def test_host_mirror_equals_restatement_on_same_scores():
    rng = np.random.default_rng(3)
    K, D, J, C = 6, 3, 2, 5
    items = list(range(40))
    occ = {i: int(rng.integers(1, 9)) for i in items if i % 5}
    ips_codes, ips_tuples = {}, {}
    for i in occ:
        codes = rng.choice(K ** D, size=C, replace=False).astype(np.int64)
        probs = np.sort(rng.random(C))[::-1].copy()
        ips_codes[i] = (codes, probs)
        ips_tuples[i] = [(dr_mstep._decode(int(c), K, D), float(p)) for c, p in zip(codes, probs)]
    got = dr_mstep.assign_paths(ips_codes, occ, items, 3, J, K, D, seed=1, penalty_factor=1e-3)
    rp = {}
    rng2 = np.random.default_rng(1)

    def random_paths(v):          # the same draws as the host mirror (seeded; the reference's are not)
        if v not in rp:
            rp[v] = [dr_mstep._decode(int(c), K, D) for c in dr_mstep._codes(rng2.integers(0, K, size=(J, D)), K)]
        return rp[v]
    # iteration t re-draws random paths in the reference too; only compare items with statistics
    want = mo.optimize(ips_tuples, occ, items, 3, J, lambda v: [], penalty_factor=1e-3)
    for i in occ:
        assert got[i] == want[i], i
    for i in items:
        assert len(got[i]) == J and all(len(p) == D and all(0 <= n < K for n in p) for p in got[i])
    assert random_paths(0)


@pytest.mark.gpu
def test_path_scores_and_assignment_vs_oracle():
    from dismember_amd import Engine
    from oracle import pyoracle as po
    K, D, L, E, n = 8, 3, 5, 16, 120
    rng = np.random.default_rng(4)
    w = synth.make_dr_model(n, K, D, L, E, rng)
    eng = Engine(0)
    eng.dr_load_model(w, E, L, K, D, n, dtype=np.float64)
    orc = po.DeepRetrieval(w, E, L, K, D, n)
    N, C, J = 300, 6, 2
    seqs = rng.integers(0, n, size=(N, L)).astype(np.int32)
    seqs[rng.random((N, L)) < 0.2] = -1
    targets = rng.integers(0, 60, size=N)                     # items 60.. never occur: random paths

    def bs(seq, beam):
        p, v = orc.beam_search(np.asarray(seq, np.int32), beam)
        return [(tuple(int(x) for x in pp), float(vv)) for pp, vv in zip(p, v)]
    samples = [(seqs[i].tolist(), int(targets[i])) for i in range(N)]
    for mode in ("batch", "streaming"):
        if mode == "batch":
            got = dr_mstep.batch_path_scores(eng, seqs, targets, C)
            want = mo.batch_path_score(samples, bs, C)
        else:
            got = dr_mstep.streaming_path_scores(eng, seqs, targets, C, 0.9)
            want = mo.streaming_path_score(samples, bs, C, 0.9, 64)
        assert set(got) == set(want)
        for item in want:
            gp = [dr_mstep._decode(int(c), K, D) for c in got[item][0]]
            assert gp == [p for p, _ in want[item]], (mode, item)
            np.testing.assert_allclose(got[item][1], [s for _, s in want[item]], rtol=1e-9)
    m = dr_mstep.optimize(eng, seqs, targets, range(n), C, J, num_iteration=2, train_mode="batch", penalty_factor=1e-4)
    want_scores = mo.batch_path_score(samples, bs, C)
    occ = {int(k): int(v) for k, v in zip(*np.unique(targets, return_counts=True))}
    wm = mo.optimize(want_scores, occ, list(range(n)), 2, J, lambda v: [], penalty_factor=1e-4)
    for item in occ:
        assert m[item] == wm[item], item
    assert all(len(m[i]) == J for i in range(n))
