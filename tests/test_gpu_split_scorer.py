"""GPU parity tests of the split-fp16 scorer mode (dm_set_scorer_mode(DM_SCORER_SPLIT_F16), include/dismember_hip.h).

Same contract as the fp32-input mode (test_gpu_parity.py): tree indices and item ids are the EXACT beam search on the
scores the device produced (oracle integer logic replayed level by level), every score within rtol 1e-4 / atol 1e-5 of
the oracle scorer.  Additionally the two device modes are compared with each other at a much tighter bound, and the
power-of-two scaling is exercised with tables far outside the fp16 range.
"""
import numpy as np
import pytest

from helpers import random_din_weights, random_histories, synthetic_tree
from test_gpu_parity import make_engine, replay_and_check

pytestmark = pytest.mark.gpu


def problem(oracle, E, depth, n_items, seed, emb_gain=1.0, w_gain=1.0):
    rng = np.random.default_rng(seed)
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    w[:NI * E] *= np.float32(emb_gain)                     # embedding table
    w[NI * E + E * E:NI * E + 3 * E * E] *= np.float32(w_gain)   # l1.W
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    odin = oracle.Din(w, E, 10, NI)
    eng = make_engine(t, w, E)
    eng.host_w = w
    seqs = random_histories(rng, t["leaf_ids"], 23, 10)
    seqs[0] = 0
    return t, otree, odin, eng, seqs


# (128, 11, 1655, 249): 498 candidates -> a 512-slot frontier that leaves room for two teams of FOUR waves only, two of
# which hold nothing but padding during the prune (regression: they must stay out of the LDS key exchange)
@pytest.mark.parametrize("E,depth,n_items,beam", [(128, 11, 1500, 50), (64, 9, 512, 16), (32, 8, 200, 100),
                                                  (128, 12, 4096, 200), (128, 11, 1655, 249)])
def test_trace_replay_split(oracle, E, depth, n_items, beam):
    t, otree, odin, eng, seqs = problem(oracle, E, depth, n_items, E * 1000 + depth)
    eng.set_scorer_mode("split_f16")
    assert eng.scorer_mode()["mode"] == "split_f16"
    replay_and_check(otree, odin, eng, seqs, beam, min(2 * beam, 200))
    replay_and_check(otree, odin, eng, seqs, beam, min(2 * beam, 200), use_mask=False)
    eng.close()


@pytest.mark.parametrize("L", [1, 3, 4, 7, 12, 16])
def test_split_all_history_lengths(oracle, L):
    rng = np.random.default_rng(50 + L)
    t = synthetic_tree(rng, 9, 400)
    NI = 1023
    w = random_din_weights(rng, 64, NI)
    otree = oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    odin = oracle.Din(w, 64, L, NI)
    eng = make_engine(t, w, 64)
    eng.set_scorer_mode("split_f16")
    seqs = random_histories(rng, t["leaf_ids"], 9, L)
    replay_and_check(otree, odin, eng, seqs, 30, 20)
    eng.close()


@pytest.mark.parametrize("emb_gain,w_gain", [(1.0, 1.0), (30.0, 1.0), (1e-6, 2e3), (7.0, 1e-2), (1.0, 3e4)])
def test_split_vs_f32_mode_and_scaling(oracle, emb_gain, w_gain):
    """The two device modes against each other: the same ids for nearly every user and scores within 2e-6 relative
    (plus 1e-7 of the largest score), for tables whose magnitudes need shifts from 2^-1 to 2^+33.  (The embedding gain
    stays moderate on the high side: attention logits grow with its square and a softmax over logits of 1e7 is
    ill-conditioned in ANY fp32 arithmetic, the oracle's included; and l1.W is not shrunk so far that every logit
    collapses onto l2.b within a few ulps, where the ranking is rounding noise in any arithmetic.)"""
    t, otree, odin, eng, _ = problem(oracle, 128, 11, 1500, 5, emb_gain, w_gain)
    rng = np.random.default_rng(9)
    seqs = random_histories(rng, t["leaf_ids"], 256, 10)
    assert eng.scorer_mode() == dict(eng.scorer_mode(), setting="auto", mode="split_f16")    # the default for E = 128
    eng.set_scorer_mode("f32")
    ids0, sc0, cnt0 = eng.tdm_beam_search(seqs, 40, 40)
    eng.set_scorer_mode("split_f16")
    ids1, sc1, cnt1 = eng.tdm_beam_search(seqs, 40, 40)
    info = eng.scorer_mode()
    assert info["mode"] == "split_f16"
    # the table's largest magnitude lands in [2^13, 2^14)
    NI = 2047
    m = np.abs(eng.host_w[:NI * 128]).max()
    assert 2.0 ** 13 <= m * 2.0 ** info["shift_emb"] < 2.0 ** 14
    assert np.array_equal(cnt0, cnt1)
    same = sum(int(np.array_equal(ids0[u, :cnt0[u]], ids1[u, :cnt1[u]])) for u in range(len(seqs)))
    assert same >= 0.9 * len(seqs), same
    big = np.abs(sc0).max()
    for u in range(len(seqs)):
        if np.array_equal(ids0[u, :cnt0[u]], ids1[u, :cnt1[u]]):
            d = np.abs(sc0[u, :cnt0[u]].astype(np.float64) - sc1[u, :cnt1[u]])
            # softmax turns an absolute error in the attention logits into a relative one in the weights: with the
            # 30x table the logits are ~900x larger, and so is the gap between ANY two fp32 evaluation orders
            tight = 2e-6 if emb_gain <= 1.0 else 1e-4
            # (a logit is a sum of terms as large as the largest logit that may cancel: the absolute part scales with it)
            assert (d <= tight * np.abs(sc0[u, :cnt0[u]]) + tight / 4 * big).all(), (u, d.max())
    # and back: the fp32 mode is unaffected by having used the other one
    eng.set_scorer_mode("f32")
    ids2, sc2, cnt2 = eng.tdm_beam_search(seqs, 40, 40)
    assert np.array_equal(ids0, ids2) and np.array_equal(sc0, sc2)
    eng.close()


def test_split_follows_weight_updates(oracle):
    """An Adam step changes the table and W1a: the fp16 planes and shifts are rebuilt before the next search."""
    t, otree, odin, eng, seqs = problem(oracle, 32, 8, 200, 3)
    eng.set_scorer_mode("split_f16")
    a = eng.tdm_beam_search(seqs, 16, 10)
    eng.train_init(lr=0.05)
    rng = np.random.default_rng(1)
    B = 256
    codes = rng.integers(0, 511, B).astype(np.int32)
    hs = rng.integers(0, 511, (B, 10)).astype(np.int32)
    eng.train_forward_backward(codes, hs, None, rng.integers(0, 2, B).astype(np.float32))
    eng.adam_step()
    w = eng.train_download("weights")
    odin2 = oracle.Din(w, 32, 10, 511)
    replay_and_check(otree, odin2, eng, seqs, 16, 10)
    b = eng.tdm_beam_search(seqs, 16, 10)
    assert not np.array_equal(a[1], b[1])
    eng.close()


def test_split_otm_mode(oracle):
    """The complete-tree (OTM) variant of the level loop under the split scorer against the fp32-input mode."""
    from dismember_amd import Engine
    rng = np.random.default_rng(12)
    depth, E = 10, 64
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    eng = Engine(0)
    eng.load_weights_din(w, E, NI)
    seqs = rng.integers((1 << depth) - 1, NI, (64, 10)).astype(np.int32)
    seqs[rng.random(seqs.shape) < 0.2] = -1
    eng.set_scorer_mode("f32")
    i0, s0, c0 = eng.otm_beam_search(seqs, 20, depth)
    eng.set_scorer_mode("split_f16")
    i1, s1, c1 = eng.otm_beam_search(seqs, 20, depth)
    assert np.array_equal(c0, c1)
    same = sum(int(np.array_equal(i0[u], i1[u])) for u in range(64))
    assert same >= 58, same
    for u in range(64):
        if np.array_equal(i0[u], i1[u]):
            np.testing.assert_allclose(s1[u, :c1[u]], s0[u, :c0[u]], rtol=2e-5, atol=2e-6)
    eng.close()


def test_split_needs_multiple_of_32(engine_fixture):
    from dismember_amd.engine import DismemberError
    with pytest.raises(DismemberError) as e:       # the bundled model has E = 16
        engine_fixture.set_scorer_mode("split_f16")
    assert e.value.code == -5                       # DM_ERR_UNSUPPORTED
    assert engine_fixture.scorer_mode()["mode"] == "f32" and engine_fixture.scorer_mode()["setting"] == "auto"


@pytest.mark.parametrize("scorer", ["auto", "f32"])
def test_repeatable_under_load(oracle, scorer):
    """4096 users (every CU busy, teams interleaving on the SIMDs) three times: bit-identical ids and scores.  A data hazard
    that the schedule only happens to avoid shows up here first ("A hazard worth writing down")."""
    t, otree, odin, eng, _ = problem(oracle, 128, 12, 4000, 41)
    eng.set_scorer_mode(scorer)
    rng = np.random.default_rng(4)
    seqs = random_histories(rng, t["leaf_ids"], 4096, 10)
    ref = eng.tdm_beam_search(seqs, 100, 100)
    for _ in range(2):
        got = eng.tdm_beam_search(seqs, 100, 100)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    # and a sample of the users against the oracle through the replay contract
    replay_and_check(otree, odin, eng, seqs[:6], 100, 100)
    eng.close()
