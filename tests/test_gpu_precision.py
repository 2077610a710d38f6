"""Pins the two claims the headline number rests on (VERDICT r1 "weak" #1, #2):

  * test_split_scorer_error_vs_exact: the split-fp16 scorer (hi + lo fp16 operands on the fp16 matrix pipe, fp32
    accumulation) is not a lower-precision result — over EVERY scored row of real searches its error against the exact
    (fp64) value of the same fp32 weights is no larger than the fp32-input MFMA kernel's, and both are no larger than the
    fp32 CPU oracle's own rounding error.  The three error figures are printed and written to gpurun_out/.
  * test_full_size_properties_depth24: the catalogue the metric is quoted on (10 M items, depth 24, E = 128, beam 200: a
    33.5 M x 128 table, past 2^32 bytes AND 2^32 elements) under pytest: TDM and OTM mode, determinism, leaf-set membership,
    scores == the general forward on rows with codes >= 2^24, the trace-replay contract against the CPU oracle on a
    user sample, one OTM training iteration (20 levels, dense Adam over 4.29 G parameters) and one JTM re-assignment step
    (levels 22 -> 24) against the oracle's aggregateWeights / reBalance.
  * test_otm_trace_exact_replay: the OTM mode's integer logic is exact — CandidateSearcher.buildBeamNodes
    (otm/.../model/CandidateSearcher.scala:109-122) replayed on the scores the GPU produced at every level, both for the
    fp32 beam kernel and for the fp64 pipeline; fp64 scores within 1e-10 / 1e-9 of the oracle's DIN[Double].
"""
import json
import os

import numpy as np
import pytest

from helpers import random_din_weights, random_histories, synthetic_tree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL, ATOL = 1e-4, 1e-5


def _engine(t, w, E):
    from dismember_amd import Engine
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"]))
    eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(w, E, (1 << (int(t["max_level"]) + 1)) - 1)
    return eng


def _traced_rows(eng, otree, seqs, beam, mode):
    """Every (code, history) row the search scored, with the score the device produced."""
    eng.set_scorer_mode(mode)
    _, _, _, tc, ts, tn = eng.tdm_beam_search_trace(seqs, beam, min(2 * beam, 200))
    codes, hist, pads, got = [], [], [], []
    for u in range(seqs.shape[0]):
        sc, mask = otree.id_to_code(seqs[u])
        m = np.zeros(sc.size, bool); m[mask] = True
        for it in range(tn.shape[1]):
            n = int(tn[u, it])
            if n:
                codes.append(tc[u, it, :n]); got.append(ts[u, it, :n])
                hist.append(np.tile(sc, (n, 1))); pads.append(np.tile(m, (n, 1)))
    codes = np.concatenate(codes); hist = np.concatenate(hist); pads = np.concatenate(pads); got = np.concatenate(got)
    return codes, hist, np.flatnonzero(pads.reshape(-1)).astype(np.int32), got


def _errs(x, exact):
    d = x.astype(np.float64) - exact
    return float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())


@pytest.mark.parametrize("E,case", [(128, "plain"), (64, "plain"), (32, "plain"), (128, "wide_range"), (128, "fp32_fallback"),
                                    (64, "wide_range")])
def test_split_scorer_error_vs_exact(oracle, E, case):
    rng = np.random.default_rng(9000 + E + len(case))
    depth, n_items, beam, U = 11, 1500, 50, 12
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI, bias_std=0.01)
    if case == "wide_range":
        # 2^12 of dynamic range INSIDE every embedding row and inside W1a's rows: the small columns live 12 bits below the
        # table's power-of-two scale, where the fp16 hi part alone would keep almost nothing
        emb = w[:NI * E].reshape(NI, E)
        emb[:, : E // 4] *= 2.0 ** -12
        emb[:, E // 4: E // 2] *= 2.0 ** -6
        l1 = w[NI * E + E * E: NI * E + 3 * E * E].reshape(E, 2 * E)
        l1[:, 1:E:3] *= 2.0 ** -11
    if case == "fp32_fallback":
        # |G + b1| above 2 max|emb| max|W1a|: this user-independent bias forces the per-user fp32-input path of the
        # attention-combine product (DESIGN.md §3 "Scaling")
        w[NI * E + 3 * E * E + 5] = 0.9
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    exact_din = oracle.Din(w.astype(np.float64), E, 10, NI)      # the exact value of the SAME fp32 weights
    o32 = oracle.Din(w, E, 10, NI)
    eng = _engine(t, w, E)
    seqs = random_histories(rng, t["leaf_ids"], U, 10)
    seqs[0] = 0
    res = {}
    for mode in ("split_f16", "f32"):
        codes, hist, pad, got = _traced_rows(eng, otree, seqs, beam, mode)
        exact = exact_din.forward(codes, hist, pad)
        ref32 = o32.forward(codes, hist, pad)
        res[mode] = dict(dev=_errs(got, exact), oracle32=_errs(ref32, exact), rows=int(codes.size),
                         max_abs_logit=float(np.abs(exact).max()))
        assert (np.abs(got - ref32) <= ATOL + RTOL * np.abs(ref32)).all()          # the stated tolerance, both arithmetics
    assert eng.scorer_mode()["setting"] == "f32"
    eng.close()
    (rs, ms), (rf, mf) = res["split_f16"]["dev"], res["f32"]["dev"]
    (ro, mo) = res["split_f16"]["oracle32"]
    line = dict(E=E, case=case, rows=res["split_f16"]["rows"], max_abs_logit=res["split_f16"]["max_abs_logit"],
                split_rms=rs, split_max=ms, f32_mfma_rms=rf, f32_mfma_max=mf, cpu_oracle_f32_rms=ro, cpu_oracle_f32_max=mo)
    print("split-vs-exact:", json.dumps(line))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "split_error.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass
    assert rs <= 1.25 * rf and ms <= 1.25 * mf + 1e-9, line       # no less accurate than the fp32-input MFMA tile
    assert rs <= ro * 1.05 and rf <= ro * 1.05, line                # and neither worse than the fp32 oracle's own rounding


@pytest.mark.parametrize("E,case", [(128, "plain"), (64, "plain"), (32, "plain"), (128, "wide_range"), (128, "big_bias")])
def test_rows_split_kernel_error_vs_exact(oracle, E, case):
    """The general-rows forward (dm_din_forward, JTM scoring) in the split-fp16 arithmetic — W1a q + (W1b att.W) c on
    v_mfma_f32_16x16x32_f16, rows_kernel.hip.inc — against the exact (fp64) value of the SAME fp32 weights: no less accurate than
    the fp32-input MFMA kernel within 25 %, neither worse than the fp32 CPU oracle's own rounding, both inside the stated tolerance."""
    from dismember_amd import Engine
    rng = np.random.default_rng(7000 + E + len(case))
    NI, B, L = 4095, 6000, 10
    w = random_din_weights(rng, E, NI, bias_std=0.01)
    if case == "wide_range":
        emb = w[:NI * E].reshape(NI, E)
        emb[:, : E // 4] *= 2.0 ** -12
        emb[:, E // 4: E // 2] *= 2.0 ** -6
        l1 = w[NI * E + E * E: NI * E + 3 * E * E].reshape(E, 2 * E)
        l1[:, 1:E:3] *= 2.0 ** -11
        l1[:, E + 2::5] *= 2.0 ** -9          # W1b columns too: M = W1b att.W inherits the range
    if case == "big_bias":
        w[NI * E + 3 * E * E: NI * E + 3 * E * E + E] = rng.normal(0, 3.0, E).astype(np.float32)    # |b1| >> |W1a q|
    codes = rng.integers(0, NI, B).astype(np.int32)
    hist = rng.integers(0, NI, (B, L)).astype(np.int32)
    hist[rng.random((B, L)) < 0.25] = -1
    hist[0] = -1
    pad = np.flatnonzero(hist.reshape(-1) < 0).astype(np.int32)
    exact = oracle.Din(w.astype(np.float64), E, L, NI).forward(codes, hist, pad)
    ref32 = oracle.Din(w, E, L, NI).forward(codes, hist, pad)
    eng = Engine(0); eng.load_weights_din(w, E, NI)
    res = {}
    for mode in ("split_f16", "f32"):
        eng.set_scorer_mode(mode)
        got = eng.din_forward(codes, hist, pad)
        assert (np.abs(got - ref32) <= ATOL + RTOL * np.abs(ref32)).all(), mode
        res[mode] = _errs(got, exact)
    eng.close()
    (rs, ms), (rf, mf), (ro, mo) = res["split_f16"], res["f32"], _errs(ref32, exact)
    line = dict(kernel="rows", E=E, case=case, rows=B, max_abs_logit=float(np.abs(exact).max()), split_rms=rs, split_max=ms,
                f32_mfma_rms=rf, f32_mfma_max=mf, cpu_oracle_f32_rms=ro, cpu_oracle_f32_max=mo)
    print("rows-split-vs-exact:", json.dumps(line))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "split_error.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass
    assert rs <= 1.25 * rf and ms <= 1.25 * mf + 1e-9, line
    assert rs <= ro * 1.05, line


# --------------------------------------------------------------------------------------------- OTM: exact integer replay
def _otm_replay(oracle, tc, ts, tn, beam, start_level, leaf_level, final_ids):
    """buildBeamNodes on the device's scores: children of every level must be exactly what the device expanded."""
    import ctypes as C
    lib = oracle.lib()
    U, levels = tn.shape
    for u in range(U):
        n0 = 1 << start_level
        ids = np.arange(n0 - 1, 2 * n0 - 1, dtype=np.int32)
        sc = np.zeros(n0, np.float64)
        for it in range(leaf_level - start_level):
            out = np.empty(2 * max(ids.size, beam), np.int32)
            n = lib.orc_otm_beam_nodes(ids.ctypes.data_as(oracle.i32p), sc.ctypes.data_as(oracle.f64p), ids.size, beam,
                                       1 if it == 0 else 0, out.ctypes.data_as(oracle.i32p))
            assert tn[u, it] == n, (u, it, tn[u, it], n)
            assert np.array_equal(tc[u, it, :n], out[:n]), (u, it)             # tree indices: bit-exact
            ids = out[:n].copy()
            sc = ts[u, it, :n].astype(np.float64)
        assert np.array_equal(final_ids[u, :ids.size], ids), u


@pytest.mark.parametrize("beam", [20, 7, 64])
def test_otm_trace_exact_replay(fixture_w64, oracle, oracle_din64, fixture_otm_mapping, beam):
    from dismember_amd import Engine
    rng = np.random.default_rng(31 + beam)
    item2node = {int(a): int(b) for a, b in fixture_otm_mapping}
    items = fixture_otm_mapping[:, 0]
    U, L, leaf_level = 24, 10, 12
    seqs = rng.choice(items, (U, L))
    seqs[:, :3][rng.random((U, 3)) < 0.4] = 0
    seqs[1] = 0
    codes = np.array([[item2node.get(int(i), -1) for i in row] for row in seqs], np.int32)
    start_level = beam.bit_length() - 1
    levels = leaf_level - start_level
    eng = Engine(0)
    eng.load_weights_din(fixture_w64, 16, 8191)
    assert eng.scorer_mode()["mode"] == "f64"                # f64 weights: the reference's arithmetic by default
    # ---- fp64 pipeline: exact replay + scores against DIN[Double] at 1e-10 / 1e-9
    ids, sc, cnt, tc, ts, tn = eng.otm_beam_search_f64(codes, beam, leaf_level, trace_levels=levels)
    assert (cnt == 2 * min(beam, 1 << (leaf_level - 1))).all() or (cnt > 0).all()
    _otm_replay(oracle, tc, ts, tn, beam, start_level, leaf_level, ids)
    n_same = 0
    for u in range(U):
        for it in range(levels):
            n = int(tn[u, it])
            pad = np.flatnonzero(np.tile(codes[u] < 0, n)).astype(np.int32)
            ref = oracle_din64.forward(tc[u, it, :n], np.tile(codes[u], (n, 1)), pad)
            assert (np.abs(ts[u, it, :n] - ref) <= 1e-10 + 1e-9 * np.abs(ref)).all(), (u, it)
        oi, osc = oracle.otm_beam_search(oracle_din64, codes[u], leaf_level, beam)
        n_same += int(np.array_equal(ids[u, :cnt[u]], oi))
        if np.array_equal(ids[u, :cnt[u]], oi):
            assert (np.abs(sc[u, :cnt[u]] - osc) <= 1e-10 + 1e-9 * np.abs(osc)).all()
    assert n_same == U            # fp64 on both sides: rounding differences of 1e-16 do not reorder a trained model's candidates
    # the float-returning entry points run the same arithmetic when f64 weights are loaded
    ids32, sc32, cnt32 = eng.otm_beam_search(codes, beam, leaf_level)
    assert np.array_equal(ids32, ids) and np.array_equal(sc32, sc.astype(np.float32))
    # ---- fp32 beam kernel (throughput mode) on the same model: exact replay on ITS scores, scores within the fp32 tolerance
    eng.set_scorer_mode("f32")
    ids_f, sc_f, cnt_f, tc_f, ts_f, tn_f = eng.otm_beam_search_trace(codes, beam, leaf_level, levels)
    _otm_replay(oracle, tc_f, ts_f, tn_f, beam, start_level, leaf_level, ids_f)
    for u in range(U):
        for it in range(levels):
            n = int(tn_f[u, it])
            pad = np.flatnonzero(np.tile(codes[u] < 0, n)).astype(np.int32)
            ref = oracle_din64.forward(tc_f[u, it, :n], np.tile(codes[u], (n, 1)), pad)
            assert (np.abs(ts_f[u, it, :n] - ref) <= ATOL + RTOL * np.abs(ref)).all(), (u, it)
    eng.close()


def test_otm_f64_synthetic_wide(oracle):
    """fp64 pipeline at E = 64 / 128 with a beam that needs the 512-slot sort and ragged padding."""
    from dismember_amd import Engine
    for E, leaf_level, beam, U in [(64, 10, 100, 9), (128, 9, 33, 5)]:
        rng = np.random.default_rng(E + beam)
        NI = (1 << (leaf_level + 1)) - 1
        w = random_din_weights(rng, E, NI, dtype=np.float64)
        eng = Engine(0)
        eng.load_weights_din(w, E, NI)
        codes = rng.integers((1 << leaf_level) - 1, NI, (U, 10)).astype(np.int32)
        codes[rng.random((U, 10)) < 0.25] = -1
        codes[0] = -1
        start_level = beam.bit_length() - 1
        levels = leaf_level - start_level
        ids, sc, cnt, tc, ts, tn = eng.otm_beam_search_f64(codes, beam, leaf_level, trace_levels=levels)
        _otm_replay(oracle, tc, ts, tn, beam, start_level, leaf_level, ids)
        odin = oracle.Din(w, E, 10, NI)
        for u in range(U):
            oi, osc = oracle.otm_beam_search(odin, codes[u], leaf_level, beam)
            assert np.array_equal(ids[u, :cnt[u]], oi), u
            assert (np.abs(sc[u, :cnt[u]] - osc) <= 1e-10 + 1e-9 * np.abs(osc)).all()
        eng.close()


# --------------------------------------------------------------------------------------------- the headline catalogue
def _host_mem_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def test_full_size_properties_depth24(oracle):
    """BASELINE configs[2]/[3] catalogue and the bench's headline workload at FULL size: 10 M items, depth 24, E = 128,
    beam 200, topk 200 (table 33 554 431 x 128 fp32 = 17.2 GB: byte and element offsets past 2^32)."""
    from dismember_amd import Engine, synth
    E, L, depth, items, beam, topk, U = 128, 10, 24, 10_000_000, 200, 200, 320
    NI = (1 << (depth + 1)) - 1
    tree = synth.make_tree(items, depth, np.random.default_rng(synth.SEED))
    eng = Engine(0)
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth)
    eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    eng.load_weights_din_synthetic(E, NI, synth.SEED, tree_depth=depth, rho=0.95)
    seqs = synth.make_users(tree["leaf_ids"], U, L, np.random.default_rng(5))
    assert eng.scorer_mode()["mode"] == "split_f16"
    ids, sc, cnt = eng.tdm_beam_search(seqs, beam, topk)
    ids2, sc2, cnt2 = eng.tdm_beam_search(seqs, beam, topk)
    assert np.array_equal(ids, ids2) and np.array_equal(sc, sc2) and np.array_equal(cnt, cnt2)      # deterministic
    lut = np.zeros(int(tree["leaf_ids"].max()) + 1, np.int32)
    lut[tree["leaf_ids"]] = tree["leaf_codes"]
    assert (cnt == topk).all() and (ids >= 1).all() and (ids <= items).all()                        # leaf-set membership
    assert all(len(set(r.tolist())) == topk for r in ids) and (np.diff(sc, axis=1) <= 0).all()
    leaf_codes = lut[ids]
    assert (leaf_codes >= (1 << 24) - 1).all() and (leaf_codes.astype(np.int64) * E * 4 >= 1 << 32).all()   # rows past 2^32 bytes
    # the beam kernel's scores == the general forward's scores of the same (leaf, history) rows (both scorer arithmetics)
    for mode in ("split_f16", "f32"):
        eng.set_scorer_mode(mode)
        idm, scm, cntm = eng.tdm_beam_search(seqs[:48], beam, topk)
        for u in (0, 17, 47):
            codes = lut[idm[u]]
            assert (codes.astype(np.int64) * E >= 1 << 31).all()                                    # element offsets past 2^31 .. 2^32
            hist, _ = eng.id_to_code(seqs[u])
            pad = np.flatnonzero(np.tile(hist < 0, topk)).astype(np.int32)
            ref = eng.din_forward(codes, np.tile(hist, (topk, 1)), pad, L=L)
            assert (np.abs(scm[u] - ref) <= ATOL + RTOL * np.abs(ref)).all(), (mode, u)
    eng.set_scorer_mode("auto")
    # OTM mode on the same table: complete depth-24 tree, 400 leaf-level candidates per user, all in the leaf range, distinct
    ocodes = np.where(seqs > 0, lut[np.clip(seqs, 0, lut.size - 1)], -1).astype(np.int32)
    oid, osc, ocnt = eng.otm_beam_search(ocodes[:64], beam, depth)
    oid2, osc2, _ = eng.otm_beam_search(ocodes[:64], beam, depth)
    assert np.array_equal(oid, oid2) and np.array_equal(osc, osc2)
    assert (ocnt == 2 * beam).all() and (oid >= (1 << depth) - 1).all() and (oid < NI).all()
    assert all(len(set(r.tolist())) == 2 * beam for r in oid)
    for u in (0, 63):
        pad = np.flatnonzero(np.tile(ocodes[u] < 0, 2 * beam)).astype(np.int32)
        ref = eng.din_forward(oid[u], np.tile(ocodes[u], (2 * beam, 1)), pad, L=L)
        assert (np.abs(osc[u] - ref) <= ATOL + RTOL * np.abs(ref)).all()
    # BASELINE configs[2] at its own size: one OTM training iteration on the complete depth-24 tree (4.29 G parameters; weights,
    # gradient and the two Adam vectors are 69 GB on the device) — pseudo targets, beam nodes and one forward/backward + dense Adam
    # per level (otm/.../optim/LocalOptimizer.scala:55-109): finite per-level losses, and the same batch trained again scores lower
    from dismember_amd.otm_train import OTMTrainer
    otr = OTMTrainer(eng, depth, 20, seq_len=L, lr=1e-3)
    trng = np.random.default_rng(12)
    first = (1 << depth) - 1
    tcodes = np.where(seqs[:6] > 0, lut[np.clip(seqs[:6], 0, lut.size - 1)], -1).astype(np.int32)
    ttargets = [(first + trng.integers(0, 1 << depth, size=2)).tolist() for _ in range(6)]
    l1 = otr.train_batch(tcodes, ttargets)
    l2 = otr.train_batch(tcodes, ttargets)
    assert len(l1) == depth - 4 and np.isfinite(l1).all() and np.isfinite(l2).all()
    assert sum(l2) < sum(l1)
    eng.load_weights_din_synthetic(E, NI, synth.SEED, tree_depth=depth, rho=0.95)          # back to the untrained table for the replay below
    # trace-replay contract against the CPU oracle on a user sample (needs a host copy of the 17.2 GB table)
    table_bytes = (NI * E + 3 * E * E + 2 * E + 1) * 4
    if _host_mem_available() < 2 * table_bytes + (8 << 30):
        eng.close()
        pytest.skip("full-size properties passed; oracle replay skipped: host has %.0f GB available" % (_host_mem_available() / 1e9))
    from test_gpu_parity import replay_and_check
    w = eng.download_weights()
    otree = oracle.TdmTree(tree["codes"], tree["ids"], tree["is_leaf"], tree["leaf_ids"], tree["leaf_codes"], depth)
    odin = oracle.Din(w, E, L, NI)
    replay_and_check(otree, odin, eng, seqs[:256], beam, topk)          # tree codes and item ids bit-exact, scores 1e-4 / 1e-5
    # each side with its own scores: identical id lists for nearly every user (a near-tie at a cut may flip)
    oids, _, ocn = otree.recommend_batch(odin, seqs[:256], topk, beam, n_threads=max(1, (os.cpu_count() or 2) - 1))
    same = sum(int(cnt[u] == ocn[u] and np.array_equal(ids[u, :cnt[u]], oids[u, :ocn[u]])) for u in range(256))
    assert same >= 0.95 * 256, same
    # BASELINE configs[3] at its own size: one JTM re-assignment step on the 10 M-item tree, the step that ends on the leaf level
    # (levels 22 -> 24: chain nodes with codes >= 2^24 - 1), for a slice of items with 4 training rows each — child weights against the
    # oracle's aggregateWeights (jtm/.../optim/TreeLearning.scala:137-174), item shards == the full matrix bit for bit, and the
    # greedy re-balance (:217-265) exact on the device's weights
    from dismember_amd.jtm import JTM
    jr = np.random.default_rng(77)
    nit = 3000
    pick = np.sort(jr.choice(tree["leaf_ids"].size, nit, replace=False))
    jitems, jcodes = tree["leaf_ids"][pick], tree["leaf_codes"][pick]
    jrows = {int(it): seqs[jr.integers(0, U, 4)].reshape(-1) for it in jitems}
    jt = JTM(eng, jitems, jcodes, depth, jrows, gap=2, seq_len=L)
    old_level, level = 22, 24
    item_node = JTM.ancestor_at_level(jt.item_code, old_level)
    w_gpu = jt.child_weights(item_node, old_level, level)
    assert w_gpu.shape == (nit, 4) and np.isfinite(w_gpu).all()
    parts = [jt.weights_range(item_node, old_level, level, a, b) for a, b in ((0, 1000), (1000, 1001), (1001, nit))]
    assert np.array_equal(np.concatenate(parts, axis=0), w_gpu)
    w_ref = oracle.jtm_child_weights(otree, odin, jt.items, jt.row_off, jt.row_ids, item_node, L, old_level, level)
    assert (np.abs(w_gpu - w_ref) <= 4 * 2 * (ATOL + RTOL * np.abs(w_ref / 4))).all()
    old_node = JTM.ancestor_at_level(jt.item_code, level)
    for node in np.unique(item_node)[:8]:
        grp = np.flatnonzero(item_node == node)
        a = jt.rebalance(w_gpu[grp], old_node[grp], int(node), old_level, level, 1)
        b = oracle.jtm_rebalance(jt.items[grp], w_gpu[grp], old_node[grp], int(node), old_level, level, 1)
        assert np.array_equal(a, b)
    # the 3 000-item sub-catalogue end to end: JTM.optimize through all twelve gap steps of the depth-24 tree on the device, against the
    # same assignment logic fed with the ORACLE's child weights (jtm/.../optim/JTM.scala:22-73, TreeLearning.scala:137-174): a bijection
    # onto leaf codes, and the same leaf for (nearly) every item (a near-tie in fp32 weights may move an item)
    proj = jt.optimize(as_array=True)
    assert np.unique(proj).size == nit and proj.min() >= (1 << depth) - 1 and proj.max() <= (1 << (depth + 1)) - 2
    ref = jt.optimize(weight_fn=lambda node, ol, lv: oracle.jtm_child_weights(otree, odin, jt.items, jt.row_off, jt.row_ids, node, L, ol, lv),
                      as_array=True)
    assert (proj == ref).mean() >= 0.9, float((proj == ref).mean())
    eng.close()


def test_full_size_otm_fp64_depth24():
    """BASELINE configs[2] in the reference's own arithmetic at its own size: a DIN[Double] over the complete depth-24 tree
    (33 554 431 x 128 doubles = 34.4 GB; with gradient and Adam state 137 GB on the device).  Size-independent properties of
    the fused fp64 beam kernel (determinism, leaf range, distinct nodes, scores == the fp64 general forward on rows whose byte
    offsets exceed 2^34), one LocalOptimizer iteration in fp64, and the f32 mirror following the trained weights."""
    from dismember_amd import Engine, synth
    from dismember_amd.otm_train import OTMTrainer
    E, L, depth, beam, U = 128, 10, 24, 200, 96
    NI = (1 << (depth + 1)) - 1
    first = (1 << depth) - 1
    eng = Engine(0)
    eng.load_weights_din_synthetic_f64(E, NI, synth.SEED)
    assert eng.scorer_mode()["mode"] == "f64"
    rng = np.random.default_rng(9)
    codes = (first + rng.integers(0, 1 << depth, size=(U, L))).astype(np.int32)
    codes[rng.random((U, L)) < 0.15] = -1
    codes[0] = -1
    ids, sc, cnt = eng.otm_beam_search_f64(codes, beam, depth)
    assert eng.last_beam_kernel().startswith("dm_beam64_kernel<128")
    ids2, sc2, cnt2 = eng.otm_beam_search_f64(codes, beam, depth)
    assert np.array_equal(ids, ids2) and np.array_equal(sc, sc2) and np.array_equal(cnt, cnt2)
    assert (cnt == 2 * beam).all() and (ids >= first).all() and (ids < NI).all()
    assert all(len(set(r.tolist())) == 2 * beam for r in ids)
    assert (ids.astype(np.int64) * E * 8 >= 1 << 34).all()
    for u in (0, 1, U - 1):
        pad = np.flatnonzero(np.tile(codes[u] < 0, 2 * beam)).astype(np.int32)
        ref = eng.din_forward(ids[u], np.tile(codes[u], (2 * beam, 1)), pad, L=L)
        assert ref.dtype == np.float64 and (np.abs(sc[u] - ref) <= 1e-10 + 1e-9 * np.abs(ref)).all(), u
    # the per-level pipeline (the fallback for frontiers that outgrow LDS) agrees with the fused kernel
    os.environ["DM_OTM64_PIPELINE"] = "1"
    try:
        idp, scp, cntp = eng.otm_beam_search_f64(codes[:8], beam, depth)
    finally:
        del os.environ["DM_OTM64_PIPELINE"]
    assert np.array_equal(idp, ids[:8]) and (np.abs(scp - sc[:8]) <= 1e-10 + 1e-9 * np.abs(sc[:8])).all()
    # one training iteration in fp64 (17 levels at beam 200 would be 8 000 rows per level with 20 users; 6 users at beam 20 here)
    otr = OTMTrainer(eng, depth, 20, seq_len=L, lr=1e-3)
    targets = [(first + rng.integers(0, 1 << depth, size=2)).tolist() for _ in range(6)]
    l1 = otr.train_batch(codes[1:7], targets)
    l2 = otr.train_batch(codes[1:7], targets)
    assert len(l1) == depth - 4 and np.isfinite(l1).all() and np.isfinite(l2).all() and sum(l2) < sum(l1)
    # the fp64 search sees the trained weights; the throughput mode (f32 mirror, rebuilt lazily) stays within the f32 tolerance of it
    ida, sca, _ = eng.otm_beam_search_f64(codes[1:3], beam, depth)
    pad = np.flatnonzero(np.tile(codes[1] < 0, 2 * beam)).astype(np.int32)
    ref = eng.din_forward(ida[0], np.tile(codes[1], (2 * beam, 1)), pad, L=L)
    assert (np.abs(sca[0] - ref) <= 1e-10 + 1e-9 * np.abs(ref)).all()
    eng.set_scorer_mode("f32")
    idf, scf, _ = eng.otm_beam_search(codes[1:3], beam, depth)
    ref32 = eng.din_forward(idf[0], np.tile(codes[1], (2 * beam, 1)), pad, L=L)
    assert (np.abs(scf[0] - ref32) <= ATOL + RTOL * np.abs(ref32)).all()
    eng.set_scorer_mode("auto")
    # ---- the conf's own batch (configs/c3_otm_10m.conf: train_batch_size 8192 users x label_num 5 targets, beam 200; round-4 verdict,
    # next #4b): 16 levels x 3.28 M candidate rows + the first level's 2.1 M through the user-grouped fp64 kernels
    # (otm/src/test/scala/OtmModelTrainSpec.scala:43-79 trains and checks that training ran; here: every level's loss finite, the
    # same batch trained again scores lower, the target lists well-formed)
    Ub, label_num = 8192, 5
    brng = np.random.default_rng(10)
    bcodes = (first + brng.integers(0, 1 << depth, size=(Ub, L))).astype(np.int32)
    bcodes[brng.random((Ub, L)) < 0.15] = -1
    btargets = (first + brng.integers(0, 1 << depth, size=(Ub, label_num))).tolist()
    otb = OTMTrainer(eng, depth, beam, seq_len=L, lr=1e-3)
    levels = depth - otb.start_level
    tgl = otb.optimal_pseudo_targets(btargets, bcodes)
    assert len(tgl) == levels
    for u in (0, 1, Ub // 2, Ub - 1):
        assert tgl[-1][u] == {int(t_): 1.0 for t_ in btargets[u]}                       # leaf level: the targets, label 1 (OTMTree.scala:38)
        for lv in range(levels - 1):
            kids, pars = tgl[lv + 1][u], tgl[lv][u]
            assert set(pars) == {(k - 1) >> 1 for k in kids} and all(0.0 <= v <= 1.0 for v in pars.values())
    lb1 = otb.train_batch(bcodes, btargets)
    st = otb.last_stats()
    assert st["rows_trained"] == Ub * (2 * (1 << otb.start_level) + (levels - 1) * 2 * beam) and st["users"] == Ub
    lb2 = otb.train_batch(bcodes, btargets)
    assert len(lb1) == levels and np.isfinite(lb1).all() and np.isfinite(lb2).all() and sum(lb2) < sum(lb1)
    # a 16-user batch against oracle/otm_oracle.py on the SAME 34 GB of weights (as trained so far): beam nodes and pseudo-target lists
    # equal, labels and the first level's loss at the fp64 contract.  Needs a host copy of the table.
    if _host_mem_available() < NI * E * 8 + (24 << 30):
        eng.close()
        pytest.skip("full-size properties passed; oracle sub-batch skipped: host has %.0f GB available" % (_host_mem_available() / 1e9))
    from oracle import otm_oracle as oo
    from oracle import pyoracle as po
    wq = eng.download_weights()
    odin = po.Din(wq, E, L, NI)
    Us = 16
    scodes, stargets = bcodes[:Us], btargets[:Us]
    got = otb.beam_search_nodes(scodes)
    refn = oo.beam_search_nodes(odin, scodes, L, otb.start_level, depth, beam)
    for lv in range(len(refn)):
        for u in range(Us):
            assert [n_ for n_, _ in got[lv][u]] == [n_ for n_, _ in refn[lv][u]], (lv, u)
    tgs = otb.optimal_pseudo_targets(stargets, scodes)
    own = oo.optimal_pseudo_targets(odin, stargets, scodes, L, otb.start_level, depth)
    for lv in range(len(own)):
        for u in range(Us):
            assert tgs[lv][u].keys() == own[lv][u].keys() and all(abs(tgs[lv][u][k] - own[lv][u][k]) < 1e-9 for k in tgs[lv][u]), (lv, u)
    c0, s0, pad0, y0 = oo.level_batch(got[0], tgs[0], scodes, L)
    x0 = odin.forward(c0, s0, pad0)
    ref_loss0 = float(np.mean(np.maximum(x0, 0) - x0 * y0 + np.log1p(np.exp(-np.abs(x0)))))
    ls = otb.train_batch(scodes, stargets)
    assert abs(ls[0] - ref_loss0) <= 1e-10 + 1e-9 * abs(ref_loss0), (ls[0], ref_loss0)
    del odin, wq
    eng.close()


def test_otm_f64_register_sort_falls_back_when_scores_share_their_top_bits(oracle):
    """The fp64 beam kernel prunes a level with a register sort on (top 53 bits of the Double.compare key, position) and redoes the
    level with the exact LDS network when two DIFFERENT scores share those bits.  A model whose output bias dwarfs everything else
    (scores = 3e4 + O(1e-9): all candidates share the prefix) must still return what the exact network returns (DM_OTM64_LDS_SORT=1)
    and replay exactly (CandidateSearcher.buildBeamNodes on the device's own scores); so must an ordinary model."""
    from dismember_amd import Engine
    rng = np.random.default_rng(99)
    E, leaf_level, beam, U, L = 32, 9, 100, 6, 10
    NI = (1 << (leaf_level + 1)) - 1
    for big_bias in (True, False):
        w = random_din_weights(rng, E, NI, dtype=np.float64, std=1e-3 if big_bias else 0.2, bias_std=0.0)
        if big_bias:
            w[-1] = 3.0e4
        codes = rng.integers((1 << leaf_level) - 1, NI, (U, L)).astype(np.int32)
        start_level = beam.bit_length() - 1
        outs = []
        for force in ("0", "1"):
            os.environ["DM_OTM64_LDS_SORT"] = force
            try:
                eng = Engine(0); eng.load_weights_din(w, E, NI)
                ids, sc, cnt, tc, ts, tn = eng.otm_beam_search_f64(codes, beam, leaf_level, trace_levels=leaf_level - start_level)
                assert eng.last_beam_kernel().startswith("dm_beam64_kernel")
                eng.close()
            finally:
                del os.environ["DM_OTM64_LDS_SORT"]
            _otm_replay(oracle, tc, ts, tn, beam, start_level, leaf_level, ids)
            outs.append((ids, sc, cnt))
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
        if big_bias:
            assert np.ptp(outs[0][1]) < 1e-6 and abs(outs[0][1].mean() - 3.0e4) < 1.0      # the scores really do share their top bits
