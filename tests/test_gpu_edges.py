"""Edge cases and full-size properties (GPU): empty batches, history lengths 1..32, extreme beams, and — at BASELINE.json's
config-2 / config-5 sizes, where the CPU oracle cannot follow — properties that do not depend on the size."""
import os

import numpy as np
import pytest

from dismember_amd import synth
from helpers import CANONICAL_TDM_QUERY, random_din_weights, random_histories, synthetic_tree

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-5


def _engine(t, w, E):
    from dismember_amd import Engine
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"]))
    eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(w, E, (1 << (int(t["max_level"]) + 1)) - 1)
    return eng


def test_empty_batches_are_not_errors():
    rng = np.random.default_rng(0)
    t = synthetic_tree(rng, 7, 100)
    eng = _engine(t, random_din_weights(rng, 16, 255), 16)
    e = np.zeros((0, 10), np.int32)
    ids, sc, cnt = eng.tdm_beam_search(e, 8, 5)
    assert ids.shape == (0, 5) and cnt.shape == (0,)
    ids, sc, cnt = eng.tdm_bruteforce_topk(e, 5)
    assert ids.shape[0] == 0
    ids, sc, cnt = eng.otm_beam_search(e, 4, 7)
    assert ids.shape[0] == 0
    eng.close()


@pytest.mark.parametrize("E,L", [(16, 1), (32, 3), (64, 12), (128, 13), (128, 16), (16, 16)])
def test_history_lengths_1_to_16(oracle, E, L):
    """L = 13..16 takes the 16-column key-fragment instantiation; positions >= L must not leak into the softmax."""
    rng = np.random.default_rng(100 * E + L)
    depth, n_items, beam = 9, 400, 24
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    odin = oracle.Din(w, E, L, NI)
    eng = _engine(t, w, E)
    seqs = random_histories(rng, t["leaf_ids"], 17, L, pad_prob=0.3)
    seqs[0] = 0                                           # all padding
    ids, sc, cnt = eng.tdm_beam_search(seqs, beam, 20)
    same = 0
    for u in range(len(seqs)):
        oi, osc = otree.recommend(odin, seqs[u], 20, beam)
        if ids[u, :cnt[u]].tolist() == oi.tolist():
            same += 1
            assert (np.abs(sc[u, :cnt[u]] - osc) <= ATOL + RTOL * np.abs(osc)).all()
    assert same >= len(seqs) - 1, same
    with pytest.raises(Exception):
        eng.tdm_beam_search(np.zeros((1, 33), np.int32), beam, 20)       # L > 32: DM_ERR_INVALID, not a wrong answer
    eng.close()


@pytest.mark.parametrize("E,L,route", [(128, 17, "fused"), (128, 24, "fused"), (128, 32, "fused"), (32, 20, "fused"), (16, 29, "fused"),
                                       (128, 24, "fused_f32"), (64, 18, "fused_f32"),
                                       (128, 17, "pipeline"), (128, 32, "pipeline"), (32, 20, "pipeline")])
def test_history_lengths_17_to_32_tdm(oracle, monkeypatch, E, L, route):
    """Histories of 17 .. 32 positions run INSIDE the fused LDS-fed kernel through a second 16-position key tile
    (dm_beam_kernel<E, 4, SPLIT, 2>, csrc/beam_kernel.hip.inc) in both scorer arithmetics; the per-level pipeline
    (csrc/tdm_pipeline.hip.inc) remains for frontiers that do not fit LDS beside two key tiles and is forced here with
    DM_LONG_PIPELINE=1.  The reference has no length limit (scalann/.../nn/Attention.scala:34-53).  Same contract as every TDM
    search: the oracle's integer logic replayed exactly on the device's per-level scores, scores within the fp32 tolerance — on a
    ragged tree, with and without the mask, and with consumed items widening the beam (Recommender.recommendItems,
    Recommender.scala:18-37)."""
    from test_gpu_parity import replay_and_check
    if route == "pipeline":
        monkeypatch.setenv("DM_LONG_PIPELINE", "1")
    else:
        monkeypatch.delenv("DM_LONG_PIPELINE", raising=False)
    rng = np.random.default_rng(1000 * E + L)
    depth, n_items, beam = 9, 400, 24
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    odin = oracle.Din(w, E, L, NI)
    eng = _engine(t, w, E)
    if route == "fused_f32":
        eng.set_scorer_mode("f32")
    seqs = random_histories(rng, t["leaf_ids"], 19, L, pad_prob=0.3)
    seqs[0] = 0                                           # all padding
    seqs[1, :] = seqs[1, 0]                               # one item repeated L times
    seqs[2, :16] = 0                                      # only positions of the second key tile are live
    seqs[3, 16:] = 0                                      # ... and only positions of the first
    for use_mask in (True, False):
        replay_and_check(otree, odin, eng, seqs, beam, 20, use_mask=use_mask)
    if route == "pipeline":
        assert "pipeline" in eng.last_beam_kernel()
    else:
        split = "true" if (route == "fused" and E % 32 == 0) else "false"
        assert eng.last_beam_kernel() == "dm_beam_kernel<%d, 4, %s, 2>" % (E, split), eng.last_beam_kernel()
    replay_and_check(otree, odin, eng, seqs[:5], 300, 50)             # a beam wider than any level of the tree
    ids, sc, cnt = eng.tdm_beam_search(seqs, beam, 20)
    same = 0
    for u in range(len(seqs)):
        oi, osc = otree.recommend(odin, seqs[u], 20, beam)
        if ids[u, :cnt[u]].tolist() == oi.tolist():
            same += 1
            assert (np.abs(sc[u, :cnt[u]] - osc) <= ATOL + RTOL * np.abs(osc)).all()
    assert same >= len(seqs) - 1, same
    # one user per call (the serving loop's entry: the host-mapped path hands over to the pipeline)
    i1, s1, c1 = eng.tdm_beam_search(seqs[3:4], beam, 20)
    assert np.array_equal(i1[0], ids[3]) and np.array_equal(s1[0], sc[3]) and c1[0] == cnt[3]
    # consumed items: dropped from the result, and a long list widens the beam
    consumed = [np.unique(rng.choice(t["leaf_ids"], 60)).tolist() if u % 2 else ids[u, :5].tolist() for u in range(len(seqs))]
    idc, scc, cntc = eng.tdm_beam_search(seqs, beam, 20, consumed=consumed, widen_consumed=True)
    agree = 0
    for u in range(len(seqs)):
        ref = otree.recommend_items(odin, seqs[u], 20, beam, consumed=consumed[u])
        assert cntc[u] == len(ref)
        assert not set(idc[u, :cntc[u]].tolist()) & set(consumed[u])
        agree += int(np.array_equal(idc[u, :cntc[u]], ref))
    assert agree >= len(seqs) - 1, agree
    eng.close()


@pytest.mark.parametrize("dtype,E,L,beam,route", [("f64", 128, 17, 33, "pipeline"), ("f64", 64, 32, 100, "pipeline"),
                                                  ("f64", 128, 17, 33, "fused"), ("f64", 64, 32, 100, "fused"), ("f64", 32, 24, 7, "fused"), ("f64", 128, 29, 300, "fused"),
                                                  ("f32", 128, 24, 33, "fused"), ("f32", 128, 32, 7, "fused"), ("f32", 32, 19, 100, "fused"),
                                                  ("f32", 128, 24, 33, "pipeline"), ("f32", 128, 32, 7, "pipeline")])
def test_history_lengths_17_to_32_otm(oracle, monkeypatch, dtype, E, L, beam, route):
    """OTM searches with 17..32 history positions: f32 models run the fused LDS-fed kernel's two-key-tile instance
    (csrc/beam_kernel.hip.inc), fp64 models the fused fp64 kernel's (dm_beam64_kernel<E, 4, 2>, csrc/beam_kernel_f64.hip.inc);
    DM_LONG_PIPELINE=1 keeps the per-level pipeline in the model's own type (csrc/otm64.hip.inc): buildBeamNodes (otm/.../model/CandidateSearcher.scala:109-122) replayed exactly on the device's scores;
    fp64 models: node lists equal to the fp64 oracle's and scores within 1e-10 / 1e-9; f32 models: scores within the fp32 tolerance."""
    from dismember_amd import Engine
    from test_gpu_precision import _otm_replay
    if route == "pipeline":
        monkeypatch.setenv("DM_LONG_PIPELINE", "1")
    else:
        monkeypatch.delenv("DM_LONG_PIPELINE", raising=False)
    leaf_level, U = 9, 7
    rng = np.random.default_rng(E + beam + L)
    NI = (1 << (leaf_level + 1)) - 1
    w64 = random_din_weights(rng, E, NI, dtype=np.float64)
    w = w64 if dtype == "f64" else w64.astype(np.float32)
    eng = Engine(0)
    eng.load_weights_din(w, E, NI)
    codes = rng.integers((1 << leaf_level) - 1, NI, (U, L)).astype(np.int32)
    codes[rng.random((U, L)) < 0.25] = -1
    codes[0] = -1
    start_level = beam.bit_length() - 1
    levels = leaf_level - start_level
    odin = oracle.Din(w.astype(np.float64), E, L, NI)
    if dtype == "f64":
        ids, sc, cnt, tc, ts, tn = eng.otm_beam_search_f64(codes, beam, leaf_level, trace_levels=levels)
        tol = (1e-10, 1e-9)
    else:
        ids, sc, cnt, tc, ts, tn = eng.otm_beam_search_trace(codes, beam, leaf_level, levels)
        tol = (ATOL, RTOL)
    if route == "pipeline":
        assert "pipeline" in eng.last_beam_kernel()
    elif dtype == "f64":
        assert eng.last_beam_kernel() == "dm_beam64_kernel<%d, 4, 2>" % E, eng.last_beam_kernel()
    else:
        assert eng.last_beam_kernel() == "dm_beam_kernel<%d, 4, true, 2>" % E, eng.last_beam_kernel()
    _otm_replay(oracle, tc, ts, tn, beam, start_level, leaf_level, ids)
    for u in range(U):
        for it in range(levels):
            n = int(tn[u, it])
            pad = np.flatnonzero(np.tile(codes[u] < 0, n)).astype(np.int32)
            ref = odin.forward(tc[u, it, :n], np.tile(codes[u], (n, 1)), pad)
            assert (np.abs(ts[u, it, :n] - ref) <= tol[0] + tol[1] * np.abs(ref)).all(), (u, it)
        if dtype == "f64":
            oi, osc = oracle.otm_beam_search(odin, codes[u], leaf_level, beam)
            assert np.array_equal(ids[u, :cnt[u]], oi), u
            assert (np.abs(sc[u, :cnt[u]] - osc) <= 1e-10 + 1e-9 * np.abs(osc)).all()
    # the plain and the device-resident entry points give the traced search's lists
    i2, s2, c2 = eng.otm_beam_search(codes, beam, leaf_level)
    assert np.array_equal(i2, ids) and np.array_equal(c2, cnt) and np.array_equal(s2, sc.astype(np.float32))
    eng.close()


def test_extreme_beams(oracle):
    rng = np.random.default_rng(5)
    depth, n_items = 8, 150
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, 32, NI)
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    odin = oracle.Din(w, 32, 10, NI)
    eng = _engine(t, w, 32)
    seqs = random_histories(rng, t["leaf_ids"], 6, 10)
    for beam, topk in [(1, 1), (1, 50), (2, 3), (3, 400), (1000, 10)]:      # beam 1: greedy descent; beam >> tree; topk >> leaves
        ids, sc, cnt = eng.tdm_beam_search(seqs, beam, topk)
        for u in range(len(seqs)):
            oi, osc = otree.recommend(odin, seqs[u], topk, beam)
            assert cnt[u] == len(oi), (beam, topk, u)
            assert (ids[u, cnt[u]:] == -1).all() or cnt[u] == topk
            if ids[u, :cnt[u]].tolist() == oi.tolist():
                assert (np.abs(sc[u, :cnt[u]] - osc) <= ATOL + RTOL * np.abs(osc)).all()
    eng.close()


def test_full_size_properties_config2():
    """BASELINE config 2 at full size (1M items, depth 20, E=128, beam 200): what must hold at any size."""
    from dismember_amd import Engine
    E, L, depth, items, beam, topk, U = 128, 10, 20, 1_000_000, 200, 200, 96
    tree = synth.make_tree(items, depth, np.random.default_rng(synth.SEED))
    eng = Engine(0)
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth)
    eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, synth.SEED, tree_depth=depth, rho=0.95)
    seqs = synth.make_users(tree["leaf_ids"], U, L, np.random.default_rng(3))
    ids, sc, cnt = eng.tdm_beam_search(seqs, beam, topk)
    ids2, sc2, cnt2 = eng.tdm_beam_search(seqs, beam, topk)
    assert np.array_equal(ids, ids2) and np.array_equal(sc, sc2) and np.array_equal(cnt, cnt2)      # deterministic
    leaf_set = set(tree["leaf_ids"].tolist())
    code_of = dict(zip(tree["leaf_ids"].tolist(), tree["leaf_codes"].tolist()))
    for u in range(U):
        r = ids[u, :cnt[u]]
        assert cnt[u] == topk and len(set(r.tolist())) == topk and set(r.tolist()) <= leaf_set
        assert (np.diff(sc[u]) <= 0).all()
    # the beam kernel's scores are the general forward's scores of the same (leaf, history) rows
    for u in (0, 17, 95):
        codes = np.array([code_of[int(i)] for i in ids[u]], np.int32)
        hist, _ = eng.id_to_code(seqs[u])
        pad = np.flatnonzero(np.tile(hist < 0, topk)).astype(np.int32)
        ref = eng.din_forward(codes, np.tile(hist, (topk, 1)), pad, L=L)
        assert (np.abs(sc[u] - ref) <= ATOL + RTOL * np.abs(ref)).all()
    # brute force bounds the beam: its k-th best score is >= the beam's k-th best, its best is >= the beam's best
    bids, bsc, bcnt = eng.tdm_bruteforce_topk(seqs[:8], topk)
    for u in range(8):
        assert (bsc[u, :topk] + ATOL + RTOL * np.abs(bsc[u, :topk]) >= sc[u]).all()
        hit = set(ids[u].tolist()) & set(bids[u].tolist())
        # every common item carries the same score on both sides
        lut = dict(zip(bids[u].tolist(), bsc[u].tolist()))
        for i, s in zip(ids[u].tolist(), sc[u].tolist()):
            if i in hit:
                assert abs(s - lut[i]) <= ATOL + RTOL * abs(lut[i])
    # a wider beam never loses the narrower beam's best item
    ids3, sc3, _ = eng.tdm_beam_search(seqs[:16], 2 * beam, topk)
    assert (sc3[:, 0] + ATOL + RTOL * np.abs(sc3[:, 0]) >= sc[:16, 0]).all()
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_full_size_properties_config5(dtype):
    """BASELINE config 5 at full size (Deep-Retrieval D=3, K=1000, beam 50, E=128, 10M items) as an f32 model and in the
    reference's own fp64 (deep-retrieval/.../model/LayerModel.scala:68-84 runs in Double).  A 256-user call takes the
    one-kernel beam search, a 1 024-user call the column-sliced pipeline (dr_sliced.hip.inc): the same users must come back
    with the same paths from both — bit for bit in fp64 (probabilities within 1e-9 relative)."""
    from dismember_amd import Engine
    E, L, K, D, items, beam, U = 128, 10, 1000, 3, 10_000_000, 50, 256
    dt = np.float32 if dtype == "f32" else np.float64
    rel = 1e-5 if dtype == "f32" else 1e-12
    eng = Engine(0)
    eng.dr_load_model_synthetic(E, L, K, D, items, synth.SEED, scale=0.05, rerank=False, dtype=dt)
    rng = np.random.default_rng(9)
    big = rng.integers(0, items, size=(4 * U, L)).astype(np.int32)
    big[rng.random((4 * U, L)) < 0.2] = -1
    big[0] = -1
    seqs = big[:U]
    p, pr, cnt = eng.dr_beam_search(seqs, beam)
    p2, pr2, _ = eng.dr_beam_search(seqs, beam)
    assert np.array_equal(p, p2) and np.array_equal(pr, pr2)
    assert (cnt == beam).all() and ((p >= 0) & (p < K)).all()
    assert (np.diff(pr, axis=1) <= 0).all() and (pr > 0).all() and (pr.sum(axis=1) <= 1 + 1e-5).all()
    codes = (p[..., 0].astype(np.int64) * K + p[..., 1]) * K + p[..., 2]
    assert all(len(set(row.tolist())) == beam for row in codes)
    # the sliced pipeline (>= 512 users per call) against the one-kernel search on the same users
    pb, prb, cb = eng.dr_beam_search(big, beam)
    assert (cb == beam).all()
    if dtype == "f64":
        # paths bit for bit; the probabilities carry the sliced pipeline's factorised softmax denominators (1e-9 relative, the
        # fp64 contract of tests/test_gpu_dr.py)
        assert np.array_equal(pb[:U], p) and np.allclose(prb[:U], pr, rtol=1e-9, atol=0)
    else:
        same = (pb[:U] == p).all(axis=(1, 2))
        assert same.mean() >= 0.9, same.mean()
        assert np.allclose(prb[:U][same], pr[same], rtol=1e-4)
    # prefix property: the top path of a wider beam is at least as probable; a beam of 1 is the greedy path
    g, gp, _ = eng.dr_beam_search(seqs[:32], 1)
    w, wp, _ = eng.dr_beam_search(seqs[:32], (4 if dtype == "f32" else 2) * beam)     # (fp64 frontiers of 256 paths outgrow the LDS)
    assert (wp[:, 0] >= pr[:32, 0] * (1 - rel)).all() and (pr[:32, 0] >= gp[:, 0] * (1 - rel)).all()
    # the beam's own best path contains the greedy first node whenever the greedy path is the best path
    same = (g[:, 0, :] == p[:32, 0, :]).all(axis=1)
    assert np.allclose(gp[same, 0], pr[:32][same, 0], rtol=max(rel, 1e-12))
    eng.close()


def test_search_without_mask(oracle):
    """useMask = false (TDM.apply for non-DIN model names, TDM.scala:26-29): padded history positions are zero rows that
    DO take part in the softmax (score 0), in the beam kernel, the brute-force mode and the general forward alike."""
    rng = np.random.default_rng(21)
    depth, n_items, beam, E, L = 9, 400, 24, 64, 10
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    odin = oracle.Din(w, E, L, NI)
    eng = _engine(t, w, E)
    seqs = random_histories(rng, t["leaf_ids"], 12, L, pad_prob=0.4)
    ids, sc, cnt = eng.tdm_beam_search(seqs, beam, 20, use_mask=False)
    idm, scm, _ = eng.tdm_beam_search(seqs, beam, 20, use_mask=True)
    assert not np.array_equal(sc, scm)                       # the flag matters on padded histories
    same = 0
    for u in range(len(seqs)):
        oi, osc = otree.recommend(odin, seqs[u], 20, beam, use_mask=False)
        if ids[u, :cnt[u]].tolist() == oi.tolist():
            same += 1
            assert (np.abs(sc[u, :cnt[u]] - osc) <= ATOL + RTOL * np.abs(osc)).all()
    assert same >= len(seqs) - 1
    bids, bsc, bcnt = eng.tdm_bruteforce_topk(seqs[:3], 10, use_mask=False)
    for u in range(3):
        hist, _ = eng.id_to_code(seqs[u])
        ref = eng.din_forward(t["leaf_codes"], np.tile(hist, (n_items, 1)), None, L=L)      # no mask indices: no masking
        order = np.argsort(-ref, kind="stable")[:10]
        assert (np.abs(bsc[u, :10] - ref[order]) <= ATOL + RTOL * np.abs(ref[order])).all()
    eng.close()


def test_jtm_rebalance_all_threads_equal_single_thread():
    """Near the root a level has a few very large parents: dm_jtm_rebalance_all gives each of them several host threads
    (parallel candidate sort, runs + stable merges).  The result must be the single-threaded dm_jtm_rebalance's, bit for bit,
    including crowded ties (weights drawn from 16 values)."""
    import ctypes as C
    from dismember_amd import Engine
    from dismember_amd import _native as N
    rng = np.random.default_rng(8)
    eng = Engine(0)
    for n_parents, n in ((1, 700_000), (3, 900_000)):
        old_level, level = (0, 2) if n_parents == 1 else (2, 4)
        nchild = 4
        parents = np.arange((1 << old_level) - 1, (1 << old_level) - 1 + n_parents, dtype=np.int32)
        item_node = parents[rng.integers(0, n_parents, n)].astype(np.int32)
        w = rng.integers(0, 16, (n, nchild)).astype(np.float32) / 4.0
        w[rng.random(n) < 0.01] = -1e6                                    # items without rows
        first_child = (item_node.astype(np.int64) << 2) + 3
        old_node = (first_child + rng.integers(0, nchild, n)).astype(np.int32)
        max_assign = int(n / n_parents / nchild * 1.02)
        out = np.empty(n, np.int32)
        i32p, f32p = N.i32p, N.f32p
        eng._chk(N.lib().dm_jtm_rebalance_all(eng._h, w.ctypes.data_as(f32p), old_node.ctypes.data_as(i32p),
                                              item_node.ctypes.data_as(i32p), n, old_level, level, max_assign, out.ctypes.data_as(i32p)))
        for pnode in parents:
            sel = np.nonzero(item_node == pnode)[0]
            ws = np.ascontiguousarray(w[sel]); on = np.ascontiguousarray(old_node[sel])
            ref = np.empty(sel.size, np.int32)
            eng._chk(N.lib().dm_jtm_rebalance(eng._h, ws.ctypes.data_as(f32p), on.ctypes.data_as(i32p), sel.size, int(pnode),
                                              old_level, level, max_assign, ref.ctypes.data_as(i32p)))
            ref = np.where(ref >= 0, ref, pnode)
            assert np.array_equal(out[sel], ref), int(pnode)
            counts = np.bincount(out[sel] - ((int(pnode) << 2) + 3), minlength=nchild)[:nchild]
            assert counts.max() <= max_assign
    eng.close()


def test_jtm_rebalance_large_parent_equals_oracle(oracle):
    """A parent with tens of thousands of items takes the selection path of the greedy re-balance (the max_assign-th element
    of the total order (moved?, -weight, list position) + a sort of the overflow only) instead of the reference's full stable
    sort: assignments must equal the oracle's reBalance (TreeLearning.scala:217-265) exactly — crowded ties (weights from 8
    values), several rounds of overflow, items without rows."""
    import ctypes as C
    from dismember_amd import Engine
    from dismember_amd import _native as N
    rng = np.random.default_rng(18)
    eng = Engine(0)
    for n, gap, slack in ((60_000, 2, 1.01), (30_000, 3, 1.10), (5_000, 2, 1.0)):
        old_level, level, node = 3, 3 + gap, 9
        nchild = 1 << gap
        w = (rng.integers(0, 8, (n, nchild)).astype(np.float32) - 3.0) / 2.0
        w[:, 0] += 1.0                                                     # a popular child: several rounds of overflow
        w[rng.random(n) < 0.02] = -1e6
        first = (node << gap) + nchild - 1
        old_node = (first + rng.integers(0, nchild, n)).astype(np.int32)
        max_assign = int(np.ceil(n / nchild * slack))
        out = np.empty(n, np.int32)
        eng._chk(N.lib().dm_jtm_rebalance(eng._h, w.ctypes.data_as(N.f32p), old_node.ctypes.data_as(N.i32p), n, node, old_level, level,
                                          max_assign, out.ctypes.data_as(N.i32p)))
        ref = oracle.jtm_rebalance(np.arange(n, dtype=np.int32), w, old_node, node, old_level, level, max_assign)
        ref = np.asarray(ref)
        assert np.array_equal(out, ref), (n, gap, int((out != ref).sum()))
    eng.close()


@pytest.mark.parametrize("case", ["root", "four_parents", "many_parents", "gap3_tight", "gap1", "nan_zero", "gap6", "gap8_root",
                                  "f64_root", "f64_parents", "f64_gap6", "f64_nan_zero"])
def test_jtm_rebalance_device_equals_host_and_oracle(oracle, case):
    """dm_jtm_rebalance_all runs every parent of a level on the device (jtm_rebalance_dev.hip.inc: rounds of stable compaction +
    one stable 64-bit radix sort on (parent | moved | descending weight key)); DM_JTM_REBALANCE=host keeps the per-parent host
    logic.  Both must give the oracle's reBalance (TreeLearning.scala:217-265) item for item: crowded ties, cascading overflow,
    capacity too small for everyone (dropped items keep their node), NaN and signed zeros in the weights."""
    import ctypes as C
    from dismember_amd import Engine
    from dismember_amd import _native as N
    rng = np.random.default_rng(len(case) * 7 + 1)
    # f64_*: double weights = OTM's TreeConstruction.reBalance (otm/.../tree/TreeConstruction.scala:304-352) through dm_otm_rebalance_all
    # (two stable sort passes on the device); gap6 / gap8: 64 / 256 children per parent (the reference's gap is free)
    cfg = dict(root=(120_000, 0, 2, 1.02), four_parents=(90_000, 2, 2, 1.0), many_parents=(200_000, 9, 2, 1.3), gap3_tight=(50_000, 3, 3, 0.8),
               gap1=(40_000, 5, 1, 1.0), nan_zero=(30_000, 1, 2, 1.05), gap6=(60_000, 2, 6, 1.1), gap8_root=(40_000, 0, 8, 1.05),
               f64_root=(80_000, 0, 2, 1.02), f64_parents=(90_000, 6, 2, 1.1), f64_gap6=(12_000, 1, 6, 0.9), f64_nan_zero=(30_000, 1, 2, 1.05))[case]
    n, old_level, gap, slack = cfg
    f64 = case.startswith("f64")
    level, C_ = old_level + gap, 1 << gap
    P = 1 << old_level
    lo = P - 1
    item_node = (lo + rng.integers(0, P, n)).astype(np.int32)
    w = (rng.integers(0, 6, (n, C_)).astype(np.float32) - 2.0) / 2.0
    w[:, 0] += 1.0
    if f64:
        w = w.astype(np.float64) + rng.integers(0, 3, (n, C_)) * 2.0 ** -40      # differences only a double holds: float keys would tie
    if case.endswith("nan_zero"):
        w[rng.random((n, C_)) < 0.05] = np.nan
        w[rng.random((n, C_)) < 0.05] = -0.0
        w[rng.random((n, C_)) < 0.05] = 0.0
    first = (item_node.astype(np.int64) << gap) + C_ - 1
    old_node = (first + rng.integers(0, C_, n)).astype(np.int32)
    old_node[rng.random(n) < 0.1] = -7                                   # items that sat elsewhere: "moved" for every child
    max_assign = max(1, int(np.ceil(n / (P * C_) * slack)))
    eng = Engine(0)
    outs = {}
    for mode in ("device", "host"):
        os.environ["DM_JTM_REBALANCE"] = mode
        try:
            out = np.empty(n, np.int32)
            fn = N.lib().dm_otm_rebalance_all if f64 else N.lib().dm_jtm_rebalance_all
            eng._chk(fn(eng._h, w.ctypes.data_as(C.POINTER(C.c_double) if f64 else N.f32p), old_node.ctypes.data_as(N.i32p), item_node.ctypes.data_as(N.i32p),
                        n, old_level, level, max_assign, out.ctypes.data_as(N.i32p)))
            outs[mode] = out
        finally:
            del os.environ["DM_JTM_REBALANCE"]
    eng.close()
    assert np.array_equal(outs["device"], outs["host"]), (case, int((outs["device"] != outs["host"]).sum()))
    ref = np.empty(n, np.int32)
    for p in np.unique(item_node):                                       # the oracle, parent by parent
        idx = np.flatnonzero(item_node == p)
        if f64:
            if p > lo + 3:                                               # the pure-Python oracle checks the first parents, host == device the rest
                ref[idx] = outs["host"][idx]
                continue
            from oracle import otm_tree_oracle as oto
            children = oto.get_children_at_level(int(p), old_level, level)
            cand = {i: oto.sort_node_weights(w[idx[i]].tolist(), children) for i in range(idx.size)}
            node_items = {}
            for i in range(idx.size):
                node_items.setdefault(cand[i][0][0], []).append((i, cand[i][0][1], 1))
            res = oto.re_balance(node_items, {i: int(old_node[idx[i]]) for i in range(idx.size)}, children, max_assign, cand)
            r = np.full(idx.size, -1, np.int64)
            for child, lst in res.items():
                for it, _, _ in lst:
                    r[it] = child
        else:
            r = np.asarray(oracle.jtm_rebalance(np.arange(idx.size, dtype=np.int32), w[idx], old_node[idx], int(p), old_level, level, max_assign))
        ref[idx] = np.where(r >= 0, r, p)
        if case == "many_parents" and p > lo + 40:
            ref[idx] = outs["host"][idx]                                 # 512 parents: the oracle checks the first 40, host == device the rest
    assert np.array_equal(outs["device"], ref), (case, int((outs["device"] != ref).sum()))
    if slack < 1.0:
        assert (outs["device"] == item_node).any()                       # somebody was dropped and kept the old node


def test_jtm_fused_device_steps_equal_separate_host_steps():
    """JTM.optimize over a 20 000-item catalogue: the whole loop as ONE call (dm_jtm_optimize_cached: projection and weights stay in
    HBM across the gap steps) and every gap step as one call (dm_jtm_step_cached) give the projection of the separate dm_jtm_child_weights_cached + host dm_jtm_rebalance_all calls, and
    it is a bijection onto the leaves (jtm/src/test/scala/JtmSpec.scala:37-51)."""
    from dismember_amd import Engine
    from dismember_amd.jtm import JTM
    items, depth, E, L, nrow = 20_000, 15, 32, 10, 3
    rng = np.random.default_rng(5)
    tree = synth.make_tree(items, depth, rng)
    eng = Engine(0)
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, 11, tree_depth=depth, rho=0.9)
    hist = synth.make_users(tree["leaf_ids"], 4096, L, np.random.default_rng(1))
    order = np.argsort(tree["leaf_ids"], kind="stable")
    pick = np.random.default_rng(2).integers(0, len(hist), size=items * nrow)
    jt = JTM.from_arrays(eng, tree["leaf_ids"][order], tree["leaf_codes"][order], depth, np.arange(items + 1, dtype=np.int64) * nrow,
                         hist[pick].reshape(-1), gap=2, seq_len=L)
    tim = {}
    fused = jt.optimize(as_array=True, timing=tim)
    assert tim["fused_step_s"] > 0 and tim["scoring_s"] > 0 and tim["rebalance_s"] > 0
    os.environ["DM_JTM_FUSED"] = "step"                        # one call per gap step (dm_jtm_step_cached), the projection through the host
    try:
        stepwise = jt.optimize(as_array=True)
    finally:
        del os.environ["DM_JTM_FUSED"]
    os.environ["DM_JTM_FUSED"] = "0"; os.environ["DM_JTM_REBALANCE"] = "host"
    try:
        sep = jt.optimize(as_array=True)
    finally:
        del os.environ["DM_JTM_FUSED"], os.environ["DM_JTM_REBALANCE"]
    eng.close()
    assert np.array_equal(fused, stepwise), int((fused != stepwise).sum())
    assert np.array_equal(fused, sep), int((fused != sep).sum())
    assert np.unique(fused).size == items and fused.min() >= (1 << depth) - 1 and fused.max() <= (1 << (depth + 1)) - 2


def test_small_searches_inside_a_training_loop_skip_the_table_rebuild(oracle):
    """AUTO mode: after an Adam step the split scorer's scale and fp16 copy of the table are stale; a small request takes the
    fp32-input kernel (reads the fp32 table as it is) instead of re-scanning the whole table, an explicit split_f16 request and a
    model that is not training keep the split kernel; results stay inside the stated tolerance of the oracle either way."""
    from dismember_amd import Engine
    from test_gpu_parity import make_engine, replay_and_check
    rng = np.random.default_rng(77)
    depth, n_items, E, beam, topk = 13, 3000, 32, 20, 10
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    eng = make_engine(t, w, E)
    seqs = random_histories(rng, t["leaf_ids"], 6, 10)
    eng.tdm_beam_search(seqs, beam, topk)
    assert eng.last_beam_kernel().startswith("dm_beam_w_kernel")
    eng.train_init(lr=1e-3)
    codes = rng.integers(0, NI, 64).astype(np.int32); hist = rng.integers(0, NI, (64, 10)).astype(np.int32)
    eng.train_forward_backward(codes, hist, None, (rng.random(64) < 0.5).astype(np.float32))
    eng.adam_step()
    w2 = eng.train_download()
    eng.tdm_beam_search(seqs, beam, topk)
    assert eng.last_beam_kernel() == "dm_beam_kernel<32, 3, false>"
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    replay_and_check(otree, oracle.Din(w2, E, 10, NI), eng, seqs, beam, topk, use_mask=True)
    eng.set_scorer_mode("split_f16")
    eng.tdm_beam_search(seqs, beam, topk)
    assert eng.last_beam_kernel().startswith("dm_beam_w_kernel")
    replay_and_check(otree, oracle.Din(w2, E, 10, NI), eng, seqs, beam, topk, use_mask=True)
    # the split copies were refreshed in the Adam step's active rows only (704 of 16 383 rows can have moved): same results as an
    # engine that loads the trained weights and splits the whole table
    assert eng.adam_last_step_rows()[1]
    got = eng.tdm_beam_search(seqs, beam, topk)
    fresh = make_engine(t, w2, E)
    fresh.set_scorer_mode("split_f16")
    want = fresh.tdm_beam_search(seqs, beam, topk)
    assert fresh.last_beam_kernel() == eng.last_beam_kernel()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    fresh.close()
    eng.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_adam_step_over_active_rows_equals_dense_stream(dtype):
    """dm_adam_step visits only the rows a gradient has ever reached (and the small matrices); for every other row the reference's
    dense update is the identity on the bits (g = s = r = 0 -> w + (-0)).  Weights AND both Adam moments after several steps, with
    rows touched in one step and left alone in the next, must equal the dense stream (DM_ADAM_DENSE=1) bit for bit; signed zeros
    in the table survive."""
    from dismember_amd import Engine
    from dismember_amd import _native as N
    E, NI, L = 32, 65535, 10
    res = {}
    for mode in ("active", "dense"):
        rng = np.random.default_rng(3)
        w = random_din_weights(rng, E, NI, dtype=dtype)
        w[5 * E: 7 * E] = -0.0; w[7 * E: 9 * E] = 0.0
        if mode == "dense":
            os.environ["DM_ADAM_DENSE"] = "1"
        try:
            eng = Engine(0); eng.load_weights_din(w, E, NI)
            eng.train_init(lr=1e-2)
            for step in range(4):
                # one gradient row per touched row, added once each: bitwise reproducible (the training kernels accumulate with
                # float atomics, whose order is not)
                rows = rng.choice(NI, 1500, replace=False).astype(np.int32)
                rows = rows[(rows != 6) & (rows != 8)]                 # the signed-zero sentinels stay untouched
                if step == 0:
                    rows[:2] = (5, 7)
                g = rng.normal(0, 1e-2, (rows.size, E)).astype(dtype)
                d_r = eng.dev_alloc(rows.nbytes); d_g = eng.dev_alloc(g.nbytes)
                eng.h2d(d_r, rows); eng.h2d(d_g, g)
                eng._chk(N.lib().dm_train_add_rows(eng._h, d_r, d_g, rows.size))
                eng.adam_step()
                eng.dev_free(d_r); eng.dev_free(d_g)
                nrows, active = eng.adam_last_step_rows()
                assert active == (mode == "active") and (nrows <= 1500 * (step + 1) if active else nrows == NI)
            res[mode] = (eng.train_download("weights"), eng.train_download("s"), eng.train_download("r"))
            eng.close()
        finally:
            os.environ.pop("DM_ADAM_DENSE", None)
    for a, b in zip(res["active"], res["dense"]):
        assert a.dtype == dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert np.signbit(res["active"][0][6 * E: 7 * E]).all() and not np.signbit(res["active"][0][8 * E: 9 * E]).any()      # untouched signed zeros


def test_jtm_cached_entry_points_reject_bad_calls():
    """dm_jtm_step_cached / dm_jtm_optimize_cached without a cached catalogue, with a catalogue of another size, with impossible
    levels: an error code and a message, never a launch."""
    import ctypes as C
    from dismember_amd import Engine
    from dismember_amd import _native as N
    lib = N.lib()
    rng = np.random.default_rng(8)
    depth, items, E, L = 9, 300, 32, 10
    tree = synth.make_tree(items, depth, rng)
    eng = Engine(0)
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, 3, tree_depth=depth, rho=0.9)
    node = np.zeros(items, np.int32); out = np.empty(items, np.int32); codes = np.sort(tree["leaf_codes"]).astype(np.int32)
    p = lambda a: a.ctypes.data_as(N.i32p)
    assert lib.dm_jtm_step_cached(eng._h, p(node), p(node), items, 0, 2, 0, 0, 1, 64, p(out)) == -3            # DM_ERR_STATE: nothing cached
    assert lib.dm_jtm_optimize_cached(eng._h, p(codes), items, depth, 2, 0, 0, 1, p(out), None) == -3
    assert b"dm_jtm_cache_rows" in lib.dm_last_error(eng._h)
    row_off = np.arange(items + 1, dtype=np.int64)
    rows = rng.choice(tree["leaf_ids"], (items, L)).astype(np.int32)
    eng._chk(lib.dm_jtm_cache_rows(eng._h, row_off.ctypes.data_as(N.i64p), p(rows), items, L))
    assert lib.dm_jtm_step_cached(eng._h, p(node), p(node), items - 1, 0, 2, 0, 0, 1, 64, p(out)) == -1        # another catalogue size
    assert lib.dm_jtm_step_cached(eng._h, p(node), p(node), items, 2, 2, 0, 0, 1, 64, p(out)) == -1            # level <= old_level
    assert lib.dm_jtm_optimize_cached(eng._h, p(codes), items, depth, 0, 0, 0, 1, p(out), None) == -1          # gap 0
    assert lib.dm_jtm_optimize_cached(eng._h, p(codes), items, depth, 2, 0, 0, 1, None, None) == -1            # no output
    secs = (C.c_double * 2)()
    eng._chk(lib.dm_jtm_optimize_cached(eng._h, p(codes), items, depth, 2, 0, 0, 1, p(out), secs))                # and a good call works
    # a handle that holds the rows of an item RANGE (a rank of a sharded run) scores that range and nothing else
    lo, hi = C.c_int64(0), C.c_int64(0)
    assert lib.dm_jtm_shard_range(items, 1, 3, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (100, 200)
    assert lib.dm_jtm_shard_range(items, 3, 3, C.byref(lo), C.byref(hi)) == -1
    full = np.empty((items, 4), np.float32)
    eng._chk(lib.dm_jtm_child_weights_cached(eng._h, p(node), 0, items, 0, 2, 0, 0, 1, full.ctypes.data_as(N.f32p)))
    lib.dm_jtm_shard_range(items, 1, 3, C.byref(lo), C.byref(hi))
    eng._chk(lib.dm_jtm_cache_rows_range(eng._h, row_off.ctypes.data_as(N.i64p), p(rows), items, L, lo.value, hi.value))
    part = np.empty((hi.value - lo.value, 4), np.float32)
    eng._chk(lib.dm_jtm_child_weights_cached(eng._h, p(node[lo.value:hi.value].copy()), lo.value, hi.value - lo.value, 0, 2, 0, 0, 1, part.ctypes.data_as(N.f32p)))
    assert np.array_equal(part, full[lo.value:hi.value])
    assert lib.dm_jtm_child_weights_cached(eng._h, p(node), 0, items, 0, 2, 0, 0, 1, full.ctypes.data_as(N.f32p)) == -3       # DM_ERR_STATE
    assert lib.dm_jtm_optimize_cached(eng._h, p(codes), items, depth, 2, 0, 0, 1, p(out), secs) == -3                          # one rank, a third of the rows
    assert lib.dm_jtm_cache_rows_range(eng._h, row_off.ctypes.data_as(N.i64p), p(rows), items, L, 5, 4) == -1
    assert np.unique(out).size == items and out.min() >= (1 << depth) - 1 and secs[0] > 0
    eng._chk(lib.dm_jtm_cache_rows(eng._h, None, None, 0, L))
    eng.close()


def test_otm_device_resident_request_equals_host_path():
    """dm_otm_beam_search_dev (request and results in HBM) == dm_otm_beam_search; codes outside the table count as padding."""
    from dismember_amd import Engine
    rng = np.random.default_rng(21)
    depth, E, L, beam, U = 9, 32, 7, 12, 50
    NI = (1 << (depth + 1)) - 1
    eng = Engine(0)
    eng.load_weights_din(random_din_weights(rng, E, NI), E, NI)
    seqs = rng.integers(0, NI, (U, L)).astype(np.int32)
    seqs[rng.random(seqs.shape) < 0.3] = -1
    ids, sc, cnt = eng.otm_beam_search(seqs, beam, depth)
    d_s = eng.dev_alloc(seqs.nbytes); d_i = eng.dev_alloc(U * 2 * beam * 4); d_v = eng.dev_alloc(U * 2 * beam * 4); d_c = eng.dev_alloc(U * 4)
    bad = seqs.copy()
    bad[3, 2] = NI + 5                       # would be an index error on the host path; padding on the device path
    ref_pad = seqs.copy(); ref_pad[3, 2] = -1
    for src, want in ((seqs, (ids, sc, cnt)), (bad, eng.otm_beam_search(ref_pad, beam, depth))):
        eng.h2d(d_s, src)
        eng.otm_beam_search_dev(d_s, U, L, beam, depth, d_i, d_v, d_c)
        eng.synchronize()
        gi = np.empty((U, 2 * beam), np.int32); gv = np.empty((U, 2 * beam), np.float32); gc = np.empty(U, np.int32)
        eng.d2h(gi, d_i); eng.d2h(gv, d_v); eng.d2h(gc, d_c)
        assert np.array_equal(gi, want[0]) and np.array_equal(gv, want[1]) and np.array_equal(gc, want[2])
    for d_ in (d_s, d_i, d_v, d_c):
        eng.dev_free(d_)
    eng.close()


def test_long_history_forward_follows_training(oracle, fixture_w32):
    """after an Adam step the general forward for L in 17..32 (din_forward_t, which reads plain transposes of
    att.W / l1.W) must see the UPDATED matrices, like every other derived copy; and gradients accumulated over several
    forward/backward calls before one Adam step keep every touched row (the touched-row list grows without losing entries)."""
    from dismember_amd import Engine
    rng = np.random.default_rng(31)
    eng = Engine(0)
    eng.load_weights_din(fixture_w32, 16, 8191)
    eng.train_init(lr=5e-2)                    # a large step: stale matrices would be far outside the tolerance
    touched = set()
    for k, B in enumerate((40, 700, 90, 1500, 64, 333)):          # growing and shrinking batches, six accumulating calls
        codes = rng.integers(1, 8191, B).astype(np.int32)
        seqs = rng.integers(0, 8191, (B, 10)).astype(np.int32)
        seqs[rng.random((B, 10)) < 0.2] = -1
        pad = np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32)
        eng.train_forward_backward(codes, seqs, pad, (rng.random(B) < 0.4).astype(np.float32))
        touched |= set(codes.tolist()) | set(seqs[seqs >= 0].tolist())
    import ctypes as C
    from dismember_amd import _native as N
    n = C.c_int64()
    eng._chk(N.lib().dm_train_export_rows(eng._h, None, None, 0, C.byref(n)))
    assert n.value == len(touched)                                  # nothing dropped, nothing duplicated
    eng.adam_step()
    w = eng.train_download("weights")
    assert np.abs(w - fixture_w32).max() > 1e-2
    for L in (20, 32, 10):
        B = 64
        codes = rng.integers(0, 8191, B).astype(np.int32)
        seqs = rng.integers(0, 8191, (B, L)).astype(np.int32)
        seqs[rng.random((B, L)) < 0.25] = -1
        pad = np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32)
        got = eng.din_forward(codes, seqs, pad)
        ref = oracle.Din(w, 16, L, 8191).forward(codes, seqs, pad)
        assert (np.abs(got - ref) <= 1e-5 + 1e-4 * np.abs(ref)).all(), L
    eng.close()


@pytest.mark.parametrize("E", [16, 32, 128])
def test_prune_order_with_mass_ties_and_signed_zeros(oracle, E):
    """The prune is a STABLE descending sort by Float.compareTo (Recommender.scala:74-87): equal scores keep frontier order,
    -0.0 sorts below +0.0.  Tables built so that whole levels tie exactly: (a) every embedding row identical -> every score of a
    level identical; (b) rows drawn from FOUR distinct vectors -> crowded ties; (c) second-layer weights and bias zero ->
    every score is +0.0 or (negative-weight variant) -0.0.  The trace-replay contract then checks the order bit for bit — this
    exercises the register sort's packed (score key, position) keys of the one-wave-per-SIMD kernel and the LDS-fed kernel alike."""
    from helpers import random_din_weights, random_histories, synthetic_tree
    from test_gpu_parity import make_engine, replay_and_check
    rng = np.random.default_rng(900 + E)
    depth, n_items = 10, 900
    NI = (1 << (depth + 1)) - 1
    t = synthetic_tree(rng, depth, n_items)
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    seqs = random_histories(rng, t["leaf_ids"], 6, 10)
    base = random_din_weights(rng, E, NI)
    emb = base[:NI * E].reshape(NI, E)
    variants = {}
    a = base.copy(); a[:NI * E].reshape(NI, E)[:] = emb[5]; variants["all_rows_equal"] = a
    b = base.copy(); b[:NI * E].reshape(NI, E)[:] = emb[rng.integers(0, 4, NI)]; variants["four_row_values"] = b
    tail0 = NI * E + E * E + 2 * E * E + E           # [emb ; att.W ; l1.W (E x 2E) ; l1.b] then l2.W [E], l2.b [1]
    c = base.copy(); c[tail0:tail0 + E + 1] = 0.0; variants["zero_output_layer"] = c
    d = base.copy(); d[tail0:tail0 + E] = -0.0; d[tail0 + E] = -0.0; variants["negative_zero_output_layer"] = d
    for name, w in variants.items():
        eng = make_engine(t, w, E)
        odin = oracle.Din(w, E, 10, NI)
        for beam, topk in ((200, 50), (37, 37), (256, 300)):
            replay_and_check(otree, odin, eng, seqs, beam, topk)
        eng.close()


def test_checkpoint_save_load_identical_recommendations(tmp_path):
    """tdm/src/test/scala/TdmModelTrainSpec.scala:85-96 (and OtmModelTrainSpec.scala:47-58): train a little, save, load into a
    fresh engine: identical weights, identical recommendations — for the f32 TDM model (tree + id maps + weights) and for an
    f64 OTM model (weights only)."""
    from dismember_amd import Engine, TDM
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    t = np.load(os.path.join(g, "tdm_tree.npz")); w = np.load(os.path.join(g, "din_f32.npy"))
    rng = np.random.default_rng(4)
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(w, 16, 8191)
    eng.train_init(lr=1e-2)
    codes = rng.integers(0, 8191, 200).astype(np.int32); seqs = rng.integers(0, 8191, (200, 10)).astype(np.int32)
    eng.train_forward_backward(codes, seqs, None, (rng.random(200) < 0.5).astype(np.float32)); eng.adam_step()
    q = np.array([0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882], np.int32)
    users = random_histories(rng, t["leaf_ids"], 32, 10)
    m = TDM(eng, "din")
    before1, before = m.recommend(q, 3, 20), eng.tdm_beam_search(users, 20, 10)
    path = str(tmp_path / "tdm_model.ck")
    m.save_model(path)
    w_trained = eng.train_download("weights")
    eng.close()
    eng2 = Engine(0)
    m2 = TDM.load_model(eng2, path, "din")
    assert eng2.E == 16 and eng2.num_index == 8191 and eng2.dtype == np.float32
    assert m2.recommend(q, 3, 20) == before1 and len(before1) == 3
    after = eng2.tdm_beam_search(users, 20, 10)
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    eng2.train_init()
    assert np.array_equal(eng2.train_download("weights"), w_trained)
    eng2.close()
    # f64 weights-only checkpoint (OTM)
    w64 = np.load(os.path.join(g, "din_f64.npy"))
    e3 = Engine(0); e3.load_weights_din(w64, 16, 8191)
    codes_o = rng.integers(4095, 8191, (5, 10)).astype(np.int32)
    b3 = e3.otm_beam_search_f64(codes_o, 20, 12)
    p3 = str(tmp_path / "otm_model.ck")
    e3.save_model(p3); e3.close()
    e4 = Engine(0); e4.load_model(p3)
    assert e4.dtype == np.float64
    a3 = e4.otm_beam_search_f64(codes_o, 20, 12)
    assert all(np.array_equal(a, b) for a, b in zip(b3, a3))
    # a file that is not a checkpoint is refused
    bad = str(tmp_path / "bad.ck"); open(bad, "wb").write(b"not a checkpoint")
    with pytest.raises(Exception):
        e4.load_model(bad)
    e4.close()

def test_checkpoint_corrupt_files_leave_the_handle_untouched(tmp_path):
    """A corrupt or truncated checkpoint returns DM_ERR_INVALID (no exception escapes the C ABI: under JNI that would abort the JVM)
    and leaves tree, id maps and weights of the handle exactly as they were; a save that fails keeps the previous file."""
    import struct
    from dismember_amd import DismemberError, Engine
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    t = np.load(os.path.join(g, "tdm_tree.npz")); w = np.load(os.path.join(g, "din_f32.npy"))
    rng = np.random.default_rng(9)
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(w, 16, 8191)
    users = random_histories(rng, t["leaf_ids"], 16, 10)
    before = eng.tdm_beam_search(users, 20, 10)
    good = str(tmp_path / "good.ck")
    eng.save_model(good)
    assert not os.path.exists(good + ".tmp")
    raw = open(good, "rb").read()
    hdr = struct.Struct("<8s8i4q")
    f = list(hdr.unpack_from(raw))
    # field indices: 1 version, 2 dtype, 3 embed, 4 max_level, 5 has_tree, 6 has_ids, 9 num_index, 10 n_elems, 11 n_nodes, 12 n_leaf_ids

    def variant(**kw):
        g_ = list(f)
        for k, v in kw.items():
            g_[{"embed": 3, "max_level": 4, "num_index": 9, "n_elems": 10, "n_nodes": 11, "n_leaf_ids": 12, "dtype": 2}[k]] = v
        return hdr.pack(*g_) + raw[hdr.size:]
    cases = {
        "huge_n_nodes": variant(n_nodes=1 << 60),               # would size a std::vector from 2^60
        "negative_leaf_ids": variant(n_leaf_ids=-5),
        "huge_num_index": variant(num_index=(1 << 62) // 16, n_elems=((1 << 62) // 16) * 16 + 3 * 256 + 33),   # num_index * E overflows
        "embed_0": variant(embed=0), "embed_4096": variant(embed=4096), "max_level_99": variant(max_level=99), "dtype_7": variant(dtype=7),
        "truncated_in_weights": raw[:len(raw) - 1000],
        "truncated_in_tree": raw[:hdr.size + 100],
        "trailing_bytes": raw + b"x" * 8,
        "bad_tree_code": raw[:hdr.size] + struct.pack("<i", -3) + raw[hdr.size + 4:],
    }
    for name, blob in cases.items():
        path = str(tmp_path / (name + ".ck"))
        open(path, "wb").write(blob)
        with pytest.raises(DismemberError) as e:
            eng.load_model(path)
        assert e.value.code == -1, (name, e.value.code)
        after = eng.tdm_beam_search(users, 20, 10)               # tree, id maps and weights are still the old ones
        assert all(np.array_equal(a, b) for a, b in zip(before, after)), name
    # a save into a directory that does not exist fails and does not disturb an existing checkpoint
    with pytest.raises(DismemberError):
        eng.save_model(str(tmp_path / "no_such_dir" / "m.ck"))
    assert open(good, "rb").read() == raw
    eng.save_model(good)                                         # overwrite in place: through good.ck.tmp + rename
    assert open(good, "rb").read() == raw and not os.path.exists(good + ".tmp")
    eng.close()



@pytest.mark.parametrize("E,dtype", [(24, np.float32), (48, np.float32), (100, np.float32), (80, np.float64), (8, np.float64)])
def test_embed_sizes_the_kernels_pad(oracle, tmp_path, E, dtype):
    """The reference takes any embedSize (S/nn/Attention.scala:34-53, T/model/DIN.scala:18-42); the kernels are built for 16 / 32 /
    64 / 128 and every other size up to 128 is zero-padded on the device, with the softmax scale 1 / sqrt(embedSize) kept the
    model's.  Forward, beam search (trace replay), one training step + Adam and the checkpoint see the MODEL's own layout."""
    from dismember_amd import Engine
    from test_gpu_parity import replay_and_check
    rng = np.random.default_rng(E)
    f64 = dtype == np.float64
    depth, items, L = 7, 100, 10
    NI = (1 << (depth + 1)) - 1
    t = synthetic_tree(rng, depth, items)
    w = random_din_weights(rng, E, NI, dtype=dtype, std=0.2, bias_std=0.2)
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], depth); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(w, E, NI)
    odin = oracle.Din(w.copy(), E, L, NI)
    B = 200
    codes = rng.integers(0, NI, B).astype(np.int32); seqs = rng.integers(0, NI, (B, L)).astype(np.int32)
    seqs[rng.random((B, L)) < 0.2] = -1
    pad = np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32)
    ref = odin.forward(codes, seqs, pad)
    got = eng.din_forward(codes, seqs, pad)
    tol = (1e-10, 1e-9) if f64 else (1e-5, 1e-4)
    assert got.dtype == dtype and (np.abs(got - ref) <= tol[0] + tol[1] * np.abs(ref)).all()
    if not f64:
        otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], depth)
        replay_and_check(otree, odin, eng, random_histories(rng, t["leaf_ids"], 6, L), 12, 8)
    else:
        oc = rng.integers((1 << depth) - 1, NI, (4, L)).astype(np.int32); oc[0, :3] = -1
        ids, sc, cnt = eng.otm_beam_search_f64(oc, 12, depth)
        for u in range(4):
            oi, osc = oracle.otm_beam_search(odin, oc[u], depth, 12)
            assert np.array_equal(ids[u, :cnt[u]], oi) and (np.abs(sc[u, :cnt[u]] - osc) <= 1e-10 + 1e-9 * np.abs(osc)).all()
    # training: gradient and Adam in the model's layout
    eng.train_init(lr=1e-3)
    y = (rng.random(B) < 0.3).astype(np.float32)
    loss = eng.train_forward_backward(codes, seqs, pad, y)
    oloss, og = odin.train_grads(codes, seqs, pad, y)
    g = eng.train_download("grad")
    assert g.shape == w.shape and abs(loss - oloss) <= (1e-10 + 1e-9 * abs(oloss) if f64 else 1e-5 + 1e-4 * abs(oloss))
    gt = (1e-10, 1e-9) if f64 else (2e-5, 1e-4)
    assert (np.abs(g - og) <= gt[0] * np.abs(og).max() + gt[1] * np.abs(og)).all()
    eng.adam_step()
    w1 = eng.train_download("weights")
    refw = w.copy(); opt = oracle.Adam(refw.size, dtype, lr=1e-3); opt.step(refw, g.copy())
    assert np.array_equal(w1, refw)
    path = str(tmp_path / "m.ck")
    eng.save_model(path); eng.close()
    e2 = Engine(0); e2.load_model(path)
    assert e2.E == E and np.array_equal(e2.din_forward(codes, seqs, pad), Engine_forward_after(w1, E, NI, codes, seqs, pad, dtype))
    e2.close()


def Engine_forward_after(w, E, NI, codes, seqs, pad, dtype):
    from dismember_amd import Engine
    e = Engine(0); e.load_weights_din(w.astype(dtype), E, NI)
    out = e.din_forward(codes, seqs, pad)
    e.close()
    return out


def test_tree_file_loaded_by_the_library_and_tdm_predict(tmp_path, fixture_tree, fixture_w32, oracle_tree, oracle_din32):
    """TDM.loadTree(treePbPath) + TDM.predict(sequence, target) (tdm/.../model/TDM.scala:10-15,50-52): the reference's own tree file
    (the bytes of its bundled data/jtm/example_tree.bin, reproduced by tree_io) parsed INSIDE the library (dm_load_tree_file) gives the
    index the array loaders give — same recommendations, same categorical sampler tables — and predict is sigmoid(Module.forward)
    of the one (history, target) row; a damaged file is refused with DM_ERR_INVALID and leaves nothing half-loaded."""
    from dismember_amd import DismemberError, Engine, TDM, tree_io
    t = fixture_tree
    leaf = t["is_leaf"] == 1
    stat = dict(zip(t["stat_ids"].tolist(), t["stat_counts"].tolist()))
    blob = tree_io.build_tree_bytes(t["ids"][leaf], t["codes"][leaf], stat)
    path = str(tmp_path / "tree.bin")
    open(path, "wb").write(blob)
    rng = np.random.default_rng(3)
    users = random_histories(rng, t["leaf_ids"], 24, 10)
    a = Engine(0)
    a.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); a.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    a.load_weights_din(fixture_w32, 16, 8191)
    b = Engine(0)
    TDM.load_tree(b, path)
    assert b.max_level == 12
    b.load_weights_din(fixture_w32, 16, 8191)
    ra, rb = a.tdm_beam_search(users, 20, 10), b.tdm_beam_search(users, 20, 10)
    assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
    # the file's node probabilities reached the sampler: sample_with_probability draws the same rows as with dm_tdm_set_node_probs
    a.set_node_probs(t["codes"], t["probs"])
    neg = np.arange(13, dtype=np.int32)
    tg = rng.choice(t["leaf_ids"], 16).astype(np.int32)
    ba = a.make_train_batch(users[:16], tg, neg, seed=5, with_prob=True)
    bb = b.make_train_batch(users[:16], tg, neg, seed=5, with_prob=True)
    assert all(np.array_equal(x, y) for x, y in zip(ba, bb))
    # predict
    m = TDM(b, "din")
    q = np.array(CANONICAL_TDM_QUERY, np.int32)
    for target in (int(t["leaf_ids"][7]), int(t["leaf_ids"][1234])):
        p = m.predict(q, target)
        codes, mask = oracle_tree.id_to_code(np.concatenate([q, [target]]).astype(np.int32))
        ref = oracle_din32.forward(codes[-1:], codes[None, :-1], mask[mask < q.size])
        assert abs(p - 1.0 / (1.0 + np.exp(-np.float64(ref[0])))) < 1e-6
    # damaged files
    for name, bad in (("cut", blob[:len(blob) - 7]), ("no_meta", blob[:blob.rindex(b"tree_meta") - 6]), ("garbage", b"\x00\x00\x00\x05hello")):
        pb = str(tmp_path / (name + ".bin"))
        open(pb, "wb").write(bad)
        with pytest.raises(DismemberError) as e:
            b.load_tree_file(pb)
        assert e.value.code == -1, name
        assert all(np.array_equal(x, y) for x, y in zip(rb, b.tdm_beam_search(users, 20, 10))), name      # the old index is intact
    a.close(); b.close()


def _free_device_bytes():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    f, t_ = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t_)) == 0
    return f.value


def test_clone_shares_weights_and_serves_concurrently(oracle):
    """dm_clone ("weights shareable read-only across handles of one device"; the reference's cloneModule() workers share
    one weight storage, tdm/.../optim/LocalOptimizer.scala:28-44, otm/src/test/scala/CloneModelSpec.scala:20-36 "cloned model shares
    weights, and a change through one is visible through the other").  A clone allocates no second table, returns what the owner
    returns — also while both search at once from two host threads — sees a training step made through the owner, refuses to load or
    train, and must be destroyed before the owner."""
    import threading
    from dismember_amd import Engine, DismemberError
    rng = np.random.default_rng(77)
    E, depth, n_items, L, beam, topk = 128, 16, 3000, 10, 64, 50          # a 67 MB table: a copied table would show
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(w, E, NI)
    seqs = random_histories(rng, t["leaf_ids"], 4096, L)
    ref = eng.tdm_beam_search(seqs, beam, topk)
    free0 = _free_device_bytes()
    c1, c2 = eng.clone(), eng.clone()
    assert free0 - _free_device_bytes() < NI * E * 4 // 2         # two clones together (streams, counters) cost less than HALF a table: nothing was copied
    for c in (c1, c2):
        got = c.tdm_beam_search(seqs, beam, topk)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))
        assert c.scorer_mode()["mode"] == eng.scorer_mode()["mode"]
    # concurrently: three host threads, three handles, one table
    outs, errs = {}, []
    def run(name, e_, lo, hi):
        try:
            for _ in range(3):
                outs[name] = e_.tdm_beam_search(seqs[lo:hi], beam, topk)
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
    ths = [threading.Thread(target=run, args=(n_, e_, lo, hi)) for n_, e_, lo, hi in (("o", eng, 0, 4096), ("a", c1, 0, 2048), ("b", c2, 2048, 4096))]
    [th.start() for th in ths]; [th.join() for th in ths]
    assert not errs, errs
    assert all(np.array_equal(a, b) for a, b in zip(outs["o"], ref))
    assert all(np.array_equal(a[:2048], b) for a, b in zip(ref, outs["a"])) and all(np.array_equal(a[2048:], b) for a, b in zip(ref, outs["b"]))
    # the other read-only entry points and the other scorer through a clone
    c1.set_scorer_mode("f32"); eng.set_scorer_mode("f32")
    assert all(np.array_equal(a, b) for a, b in zip(eng.tdm_beam_search(seqs[:256], beam, topk), c1.tdm_beam_search(seqs[:256], beam, topk)))
    eng.set_scorer_mode("auto"); c1.set_scorer_mode("auto")
    codes = rng.integers(0, NI, 300).astype(np.int32); hs = rng.integers(0, NI, (300, L)).astype(np.int32)
    assert np.array_equal(eng.din_forward(codes, hs, None), c2.din_forward(codes, hs, None))
    assert np.array_equal(eng.id_to_code(seqs[0])[0], c2.id_to_code(seqs[0])[0])
    # training through the owner is visible through the clones (their next call re-mirrors the refreshed copies)
    eng.train_init(lr=1e-2)
    y = (rng.random(300) < 0.3).astype(np.float32)
    eng.train_forward_backward(codes, hs, None, y); eng.adam_step(1.0)
    after = eng.tdm_beam_search(seqs[:512], beam, topk)
    assert not all(np.array_equal(a[:512], b) for a, b in zip(ref, after))          # the step moved the scores
    for c in (c1, c2):
        got = c.tdm_beam_search(seqs[:512], beam, topk)
        assert all(np.array_equal(a, b) for a, b in zip(after, got))
    assert np.array_equal(eng.din_forward(codes, hs, None), c1.din_forward(codes, hs, None))
    # a clone neither loads nor trains; the owner outlives its clones
    with pytest.raises(DismemberError):
        c1.load_weights_din(w, E, NI)
    with pytest.raises(DismemberError):
        c1.train_init(lr=1e-3)
    with pytest.raises(DismemberError):
        eng.close()
    c3 = c1.clone()                                                # a clone of a clone shares the same owner
    assert all(np.array_equal(a, b) for a, b in zip(after, c3.tdm_beam_search(seqs[:512], beam, topk)))
    c3.close(); c1.close(); c2.close()
    eng.close()


def test_clone_of_an_f64_model_serves_the_fp64_search(oracle):
    """The same for a DIN[Double] (OTM): the clone reads the owner's fp64 table and fragments; node lists and scores identical."""
    from dismember_amd import Engine
    rng = np.random.default_rng(78)
    E, leaf_level, beam = 32, 9, 24
    NI = (1 << (leaf_level + 1)) - 1
    w = random_din_weights(rng, E, NI, dtype=np.float64, std=0.2, bias_std=0.1)
    eng = Engine(0)
    eng.load_weights_din(w, E, NI)
    codes = rng.integers((1 << leaf_level) - 1, NI, (64, 10)).astype(np.int32)
    codes[rng.random(codes.shape) < 0.2] = -1
    c = eng.clone()
    a = eng.otm_beam_search_f64(codes, beam, leaf_level)
    b = c.otm_beam_search_f64(codes, beam, leaf_level)
    assert all(np.array_equal(x, y_) for x, y_ in zip(a, b))
    din = oracle.Din(w, E, 10, NI)
    oi, osc = oracle.otm_beam_search(din, codes[3], leaf_level, beam)
    assert np.array_equal(b[0][3, :b[2][3]], oi)
    c.close(); eng.close()


def test_host_buffer_searches_pipeline_their_downloads_and_equal_the_device_resident_path():
    """Large host-buffer requests (dm_tdm_beam_search / dm_otm_beam_search) are cut into chunks of users whose result downloads run
    under the kernels of the chunks behind them (host_pipe_chunks, dm_hip.hip).  Same ids, scores and counts as ONE launch over the
    device-resident request (dm_*_beam_search_dev), for a user count that does not divide into the chunks; the scored-rows counter
    covers the whole request."""
    from dismember_amd import Engine
    rng = np.random.default_rng(91)
    E, depth, n_items, L, beam, topk = 32, 11, 2000, 10, 100, 200
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(random_din_weights(rng, E, NI), E, NI)
    U = 50_001                                                     # 80 MB of results: three chunks, the last one ragged
    seqs = random_histories(rng, t["leaf_ids"], U, L)
    d_seq, d_ids, d_sc, d_cnt = eng.dev_alloc(U * L * 4), eng.dev_alloc(U * topk * 4), eng.dev_alloc(U * topk * 4), eng.dev_alloc(U * 4)
    eng.h2d(d_seq, seqs)
    eng.tdm_beam_search_dev(d_seq, U, L, beam, topk, d_ids, d_sc, d_cnt)
    eng.synchronize()
    rows_dev = eng.last_scored_rows()
    ids, sc, cnt = np.empty((U, topk), np.int32), np.empty((U, topk), np.float32), np.empty(U, np.int32)
    eng.d2h(ids, d_ids); eng.d2h(sc, d_sc); eng.d2h(cnt, d_cnt)
    out = (np.full((U, topk), -7, np.int32), np.full((U, topk), -7, np.float32), np.full(U, -7, np.int32))
    eng.tdm_beam_search(seqs, beam, topk, out=out)
    assert np.array_equal(out[0], ids) and np.array_equal(out[1], sc) and np.array_equal(out[2], cnt)
    assert eng.last_scored_rows() == rows_dev
    # OTM mode, same table
    first = (1 << depth) - 1
    codes = (first + rng.integers(0, 1 << depth, (U, L))).astype(np.int32)
    codes[rng.random((U, L)) < 0.2] = -1
    eng.h2d(d_seq, codes)
    eng.otm_beam_search_dev(d_seq, U, L, beam, depth, d_ids, d_sc, d_cnt)
    eng.synchronize()
    eng.d2h(ids, d_ids); eng.d2h(sc, d_sc); eng.d2h(cnt, d_cnt)
    o2 = eng.otm_beam_search(codes, beam, depth)
    assert np.array_equal(o2[0], ids) and np.array_equal(o2[1], sc) and np.array_equal(o2[2], cnt)
    for d_ in (d_seq, d_ids, d_sc, d_cnt):
        eng.dev_free(d_)
    eng.close()
