"""GPU parity tests of the Deep-Retrieval path (row A13): dm_dr_* through the C ABI against oracle/dr_body.inc.

Contract:
  * fp64 model (the reference's arithmetic type): paths BIT-EXACT against the fp64 oracle, probabilities / rerank
    logits within rtol 1e-9 (summation order differs: MFMA GEMM + precomputed node tables vs sequential dot).
  * fp32 model (throughput mode): every returned path's probability within rtol 1e-4 of the oracle's probability
    OF THAT PATH, probabilities non-increasing, and the path lists identical to the fp64 oracle's for nearly all
    users (the cut is a discontinuous function of the probabilities).
  * ties (equal probabilities): exact stable order (earlier parent path, then lower node) in both precisions.
"""
import os

import numpy as np
import pytest

from dismember_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(autouse=True, params=["one_kernel", "sliced"])
def dr_search_path(request):
    """Every test runs twice: with the one-workgroup-per-user kernel (what a batch below 512 users takes) and with the column-sliced
    pipeline of dr_sliced.hip.inc forced onto these small batches (DM_DR_SLICED_MIN_USERS is read at model load)."""
    if request.param == "sliced":
        os.environ["DM_DR_SLICED_MIN_USERS"] = "1"
    yield request.param
    os.environ.pop("DM_DR_SLICED_MIN_USERS", None)


def make(K, D, L, E, num_item, seed, dtype, scale=0.3, J=2, with_paths=True, collapse=False):
    from dismember_amd import Engine
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed)
    w = synth.make_dr_model(num_item, K, D, L, E, rng, scale=scale)
    if dtype == np.float32:     # the oracle sees exactly the values the device holds
        for k, v in list(w.items()):
            w[k] = [a.astype(np.float32).astype(np.float64) for a in v] if isinstance(v, list) else v.astype(np.float32).astype(np.float64)
    eng = Engine(0)
    eng.dr_load_model(w, E, L, K, D, num_item, dtype=dtype)
    pi = None
    if with_paths:
        pi = synth.dr_path_items(synth.make_dr_paths(num_item, K, D, J, rng), collapse=collapse)
        eng.dr_load_path_items(*pi)
    orc = po.DeepRetrieval(w, E, L, K, D, num_item, path_items=pi)
    return eng, orc, w, rng


def histories(rng, U, L, num_item, pad_p=0.2):
    s = rng.integers(0, num_item, size=(U, L)).astype(np.int32)
    s[rng.random((U, L)) < pad_p] = -1
    s[0, :] = -1                      # an all-padding history
    return s


def path_prob(orc, seq, path):
    from oracle import pyoracle as po
    ids, pr = list(seq), 1.0
    for d, node in enumerate(path):
        pr *= po.dr_softmax(orc.inference(ids, d))[node]
        ids.append(int(node) + orc.num_item + d * orc.K)
    return pr


@pytest.mark.parametrize("K,D,L,E,beam", [(7, 3, 4, 16, 5), (7, 3, 4, 16, 20), (100, 3, 10, 16, 20), (65, 2, 3, 32, 1),
                                          (130, 4, 5, 16, 33), (300, 3, 10, 64, 50)])
def test_beam_search_f64_exact(K, D, L, E, beam):
    eng, orc, w, rng = make(K, D, L, E, 500, 11, np.float64, with_paths=False)
    seqs = histories(rng, 24, L, 500)
    paths, probs, cnt = eng.dr_beam_search(seqs, beam)
    for u in range(len(seqs)):
        op, ov = orc.beam_search(seqs[u], beam)
        assert cnt[u] == len(op)
        assert paths[u, :cnt[u]].tolist() == op.tolist(), u
        np.testing.assert_allclose(probs[u, :cnt[u]], ov, rtol=1e-9, atol=0)
        assert (paths[u, cnt[u]:] == -1).all()


def test_beam_search_c5_shape_f64():
    """BASELINE config 5 shape (D=3, K=1000, beam=50, E=128, L=10) on a small catalogue."""
    eng, orc, w, rng = make(1000, 3, 10, 128, 3000, 5, np.float64, scale=0.05, with_paths=False)
    seqs = histories(rng, 6, 10, 3000)
    paths, probs, cnt = eng.dr_beam_search(seqs, 50)
    for u in range(len(seqs)):
        op, ov = orc.beam_search(seqs[u], 50)
        assert paths[u, :cnt[u]].tolist() == op.tolist(), u
        np.testing.assert_allclose(probs[u, :cnt[u]], ov, rtol=1e-9)


@pytest.mark.parametrize("K,D,L,E,beam", [(100, 3, 10, 16, 20), (1000, 3, 10, 128, 50)])
def test_beam_search_f32(K, D, L, E, beam):
    scale = 0.05 if K == 1000 else 0.3
    eng, orc, w, rng = make(K, D, L, E, 2000, 7, np.float32, scale=scale, with_paths=False)
    U = 12 if K == 1000 else 48
    seqs = histories(rng, U, L, 2000)
    paths, probs, cnt = eng.dr_beam_search(seqs, beam)
    same = 0
    for u in range(U):
        assert cnt[u] == beam
        assert (np.diff(probs[u]) <= 0).all()
        assert len({tuple(p) for p in paths[u]}) == beam
        for q in (0, 1, beam // 2, beam - 1):
            assert abs(path_prob(orc, seqs[u], paths[u, q]) - probs[u, q]) <= 1e-4 * probs[u, q] + 1e-30
        op, ov = orc.beam_search(seqs[u], beam)
        same += int(paths[u].tolist() == op.tolist())
        # the probability mass the beam keeps is the same up to rounding even when a near-tie flips
        assert abs(probs[u].sum() - ov.sum()) <= 1e-4 * ov.sum()
    assert same >= 0.9 * U, same


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ties_are_stable(dtype):
    from dismember_amd import Engine
    K, D, L, E, n = 70, 3, 3, 16, 10
    w = synth.make_dr_model(n, K, D, L, E, np.random.default_rng(0))
    for k in ("layer_w", "layer_b"):
        w[k] = [np.zeros_like(a) for a in w[k]]
    eng = Engine(0)
    eng.dr_load_model(w, E, L, K, D, n, dtype=dtype)
    for beam in (7, 100):
        paths, probs, cnt = eng.dr_beam_search(np.array([[1, 2, 3]], np.int32), beam)
        # layer 0 keeps nodes 0..min(beam,K)-1; layer 1 keeps (0, 0..), and so on: lexicographic prefixes
        exp = []
        l0 = list(range(min(beam, K)))
        l1 = [(a, b) for a in l0 for b in range(K)][:beam]
        l2 = [(a, b, c) for (a, b) in l1 for c in range(K)][:beam]
        exp = [list(t) for t in l2]
        assert cnt[0] == beam and paths[0].tolist() == exp
        np.testing.assert_allclose(probs[0], 1.0 / K ** 3, rtol=1e-6)


@pytest.mark.parametrize("dtype,collapse", [(np.float64, False), (np.float64, True), (np.float32, False)])
def test_recommend_vs_oracle(dtype, collapse):
    K, D, L, E, n = 12, 3, 6, 32, 4000        # 1728 paths for 8000 (item, path) pairs: crowded buckets
    eng, orc, w, rng = make(K, D, L, E, n, 21, dtype, collapse=collapse)
    seqs = histories(rng, 40, L, n)
    ids, sc, cnt = eng.dr_recommend(seqs, 25, 30)
    same = 0
    for u in range(len(seqs)):
        oi, osc = orc.recommend(seqs[u], 30, 25)
        if dtype == np.float64:
            assert cnt[u] == len(oi) and ids[u, :cnt[u]].tolist() == oi.tolist(), u
            np.testing.assert_allclose(sc[u, :cnt[u]], osc, rtol=1e-9, atol=1e-12)
        else:
            assert cnt[u] == len(oi)
            same += int(ids[u, :cnt[u]].tolist() == oi.tolist())
            if ids[u, :cnt[u]].tolist() == oi.tolist():
                np.testing.assert_allclose(sc[u, :cnt[u]], osc, rtol=1e-4, atol=1e-5)
            assert (np.diff(sc[u, :cnt[u]]) <= 0).all()
        assert (ids[u, cnt[u]:] == -1).all()
    if dtype == np.float32:
        assert same >= 0.85 * len(seqs), same


def test_recommend_bundled_mapping_and_facade():
    """data/dr/example_mapping.bin (3325 items, 2 paths of 3 nodes < 100 each) with the query of
    DeepRetrievalSpec.scala:110; synthetic weights (the reference ships no DR model)."""
    from dismember_amd import Engine
    from dismember_amd.facade import DeepRetrieval
    from oracle import pyoracle as po
    d = np.load(os.path.join(GOLDEN, "dr_mapping.npz"))
    n, K, D, L, E = 3325, 100, 3, 10, 16
    ip = np.empty_like(d["paths"])
    ip[d["ids"]] = d["paths"]
    pi = synth.dr_path_items(ip)
    w = synth.make_dr_model(n, K, D, L, E, np.random.default_rng(2))
    eng = Engine(0)
    eng.dr_load_model(w, E, L, K, D, n, dtype=np.float64)
    eng.dr_load_path_items(*pi)
    orc = po.DeepRetrieval(w, E, L, K, D, n, path_items=pi)
    item_id = {int(i): int(j) for i, j in zip(d["items"], d["ids"])}
    id_item = {v: k for k, v in item_id.items()}
    model = DeepRetrieval(eng, item_id)
    query = [1, 2, 3, 4, 5, 6, 7, 89, 2628, 1681]
    # random weights know nothing about the 6 627 occupied paths (of 10^6): a wide beam is needed to hit some
    recs = model.recommend(query, 10, 1000)
    seq_ids = [item_id.get(i, -1) for i in query]
    oi, osc = orc.recommend(seq_ids, 10, 1000)
    assert len(recs) >= 3                               # DeepRetrievalSpec.scala:112
    assert [r[0] for r in recs] == [id_item[int(i)] for i in oi]
    np.testing.assert_allclose([r[1] for r in recs], 1.0 / (1.0 + np.exp(-osc)), rtol=1e-9)


def test_errors():
    from dismember_amd import Engine
    from dismember_amd.engine import DismemberError
    eng = Engine(0)
    eng.dr_dims = dict(E=16, L=4, K=7, D=3, num_item=50, dtype=np.dtype(np.float64))
    with pytest.raises(DismemberError) as e:
        eng.dr_beam_search(np.zeros((1, 4), np.int32), 5)
    assert e.value.code == -3                           # DM_ERR_STATE
    eng, orc, w, rng = make(7, 3, 4, 16, 50, 3, np.float64, with_paths=False)
    with pytest.raises(DismemberError) as e:
        eng.dr_beam_search(np.array([[0, 1, 50, 2]], np.int32), 5)
    assert e.value.code == -4                           # DM_ERR_INDEX
    with pytest.raises(DismemberError) as e:
        eng.dr_recommend(np.array([[0, 1, 2, 3]], np.int32), 5, 5)
    assert e.value.code == -3                           # no path table
    with pytest.raises(DismemberError) as e:
        eng.dr_load_path_items(np.array([[1, 2, 3], [1, 2, 3]], np.int32), [0, 1, 2], [4, 5])
    assert e.value.code == -1                           # duplicate path


def test_split_history_gemm_vs_fp32_gemm():
    """Float models with E % 32 == 0 run the history GEMM on the fp16 matrix pipe (hi + lo operand split, fp32 accumulation:
    dm_set_scorer_mode).  Against the fp32-input GEMM of the same model: the same paths for (nearly) every
    user, probabilities within 2e-5 relative; and against the fp64 oracle the usual f32 contract."""
    K, D, L, E, beam, n = 200, 3, 10, 64, 30, 5000
    eng, orc, w, rng = make(K, D, L, E, n, 17, np.float32, scale=0.1, with_paths=False)
    seqs = histories(rng, 96, L, n)
    eng.set_scorer_mode("f32")
    p0, v0, c0 = eng.dr_beam_search(seqs, beam)
    eng.set_scorer_mode("auto")
    p1, v1, c1 = eng.dr_beam_search(seqs, beam)
    assert np.array_equal(c0, c1)
    same = 0
    for u in range(len(seqs)):
        if np.array_equal(p0[u], p1[u]):
            same += 1
            np.testing.assert_allclose(v1[u], v0[u], rtol=2e-5, atol=0)
        for q in (0, beam - 1):
            assert abs(path_prob(orc, seqs[u], p1[u, q]) - v1[u, q]) <= 1e-4 * v1[u, q] + 1e-30
    assert same >= 0.95 * len(seqs), same
    assert not np.array_equal(v0, v1)        # the two GEMMs really are different kernels


def test_split_scorer_by_name_is_refused_where_the_split_gemm_does_not_exist():
    """E = 32 (not a multiple of 64): no split planes are built, DM_SCORER_AUTO runs the fp32 GEMM — and asking for SPLIT_F16 by name is
    DM_ERR_UNSUPPORTED, not a silent fp32 run; E = 1088 (> 1024) pads histories with a zero block as long as a row's records."""
    from dismember_amd.engine import DismemberError
    eng, orc, w, rng = make(50, 2, 6, 32, 400, 19, np.float32, scale=0.1, with_paths=False)
    seqs = histories(rng, 8, 6, 400)
    eng.set_scorer_mode("auto")
    p0, v0, c0 = eng.dr_beam_search(seqs, 10)
    eng.set_scorer_mode("split_f16")
    with pytest.raises(DismemberError) as e:
        eng.dr_beam_search(seqs, 10)
    assert e.value.code == -5
    eng.set_scorer_mode("f32")
    p1, v1, c1 = eng.dr_beam_search(seqs, 10)
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    eng.close()
    K, D, L, E, n = 32, 2, 4, 1088, 300
    eng, orc, w, rng = make(K, D, L, E, n, 23, np.float32, scale=0.02, with_paths=False)
    seqs = histories(rng, 40, L, n)
    eng.set_scorer_mode("f32")
    p0, v0, c0 = eng.dr_beam_search(seqs, 8)
    eng.set_scorer_mode("auto")
    p1, v1, c1 = eng.dr_beam_search(seqs, 8)
    same = sum(np.array_equal(p0[u], p1[u]) for u in range(len(seqs)))
    assert same >= 0.9 * len(seqs), same
    for u in range(len(seqs)):
        if np.array_equal(p0[u], p1[u]):
            np.testing.assert_allclose(v1[u], v0[u], rtol=5e-5, atol=0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sliced_pipeline_equals_one_kernel_on_a_batch(dtype, dr_search_path):
    """600 users (above the default switch-over of 512), BASELINE config 5's shape: the column-sliced pipeline — history factors and
    tabulated table factors instead of one exp per candidate, one-wave cuts — returns the paths of dr_beam_kernel for every user and
    its probabilities to rounding (fp64: 1e-12; the two differ in the summation order of the softmax denominators only)."""
    if dr_search_path == "sliced":
        pytest.skip("the batch takes the sliced pipeline by its size; the forced variant adds nothing")
    from dismember_amd import Engine
    rng = np.random.default_rng(21)
    K, D, L, E, n, beam, U = 1000, 3, 10, 32, 5000, 50, 600
    w = synth.make_dr_model(n, K, D, L, E, rng, scale=0.05)
    seqs = histories(rng, U, L, n)
    seqs[1] = seqs[2]                                   # identical users: identical results
    out = {}
    for mode in ("sliced", "one_kernel"):
        if mode == "one_kernel":
            os.environ["DM_DR_SLICED"] = "0"
        try:
            eng = Engine(0)
            eng.dr_load_model(w, E, L, K, D, n, dtype=dtype)
            out[mode] = eng.dr_beam_search(seqs, beam)
            eng.close()
        finally:
            os.environ.pop("DM_DR_SLICED", None)
    (pa, pra, ca), (pb, prb, cb) = out["sliced"], out["one_kernel"]
    assert np.array_equal(ca, cb) and (ca == beam).all()
    same = (pa == pb).all(axis=(1, 2))
    if dtype == np.float64:
        assert same.all(), int((~same).sum())
        np.testing.assert_allclose(pra, prb, rtol=1e-12, atol=0)
    else:
        assert same.mean() > 0.9                          # f32: near-ties at the cut may flip between two summation orders
        np.testing.assert_allclose(pra[same], prb[same], rtol=2e-5, atol=0)
    assert np.array_equal(pa[1], pa[2]) and np.array_equal(pra[1], pra[2])
    assert (np.diff(pra, axis=1) <= 0).all()


@pytest.mark.parametrize("K,L,E,U", [(100, 13, 64, 600), (86, 3, 128, 1061), (1000, 10, 128, 12288 + 37)])
def test_presplit_history_gemm_is_bit_identical_to_the_128_tile_kernel(K, L, E, U, dr_search_path):
    """Batches that fill whole rounds of 256 x 256 tiles (dr_gemm_x_pays) run the history GEMM on them, over operands split into fp16 hi/lo records once at model
    load and copied global -> LDS directly (dr_gemm_split_x_kernel).  Same split arithmetic and the same order of MFMAs per accumulator
    as the 128 x 128 kernel, so the two must agree in every bit: paths, probabilities and counts of a whole search, on shapes that leave
    ragged row and column tiles, idle XCD slots, E = 64 (two stages per history position) and padding ids."""
    if dr_search_path == "sliced" and U > 2000:
        pytest.skip("the batch takes the sliced pipeline by its size; the forced variant adds nothing")
    from dismember_amd import Engine
    rng = np.random.default_rng(33)
    D, n, beam = 3, 5000, 10
    w = synth.make_dr_model(n, K, D, L, E, rng, scale=0.08)
    seqs = histories(rng, U, L, n)
    out = {}
    for x in ("0", "1"):
        os.environ["DM_DR_GEMM_X"] = x
        os.environ["DM_DR_GEMM_X_MIN_ROWS"] = "1" if U < 12288 else "12288"
        try:
            eng = Engine(0)
            eng.dr_load_model(w, E, L, K, D, n, dtype=np.float32)
            out[x] = eng.dr_beam_search(seqs, beam)
            eng.set_scorer_mode("f32")
            ref = eng.dr_beam_search(seqs[:64], beam)
            eng.close()
        finally:
            os.environ.pop("DM_DR_GEMM_X", None)
            os.environ.pop("DM_DR_GEMM_X_MIN_ROWS", None)
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)
    assert (out["1"][2] > 0).all()
    same = (out["1"][0][:64] == ref[0]).all(axis=(1, 2))          # and close to the fp32-input GEMM, like the 128 x 128 kernel
    assert same.mean() > 0.9
    np.testing.assert_allclose(out["1"][1][:64][same], ref[1][same], rtol=2e-5, atol=0)
