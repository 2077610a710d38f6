"""Evaluator + metrics: known answers for the oracle restatement and the host mirror (CPU), and
the device evaluator against the oracle evaluator on the reference's bundled tree + model (GPU)."""
import math
import os

import numpy as np
import pytest

from dismember_amd import evaluation as ev
from oracle import eval_oracle as eo


def test_metrics_known_answers():
    # hits at positions 0 and 2 of 4 returned, 3 labels: precision 2/4, recall 2/3,
    # dcg = 1 + log2/log4 = 1.5, idcg = 1 + log2/log3
    rec, lab = [7, 1, 9, 4], [9, 7, 100]
    exp = (0.5, 2 / 3.0, 1.5 / (1 + math.log(2) / math.log(3)))
    for f in (eo.compute_metrics, ev.compute_metrics):
        got = f(rec, lab)
        assert got == pytest.approx(exp, rel=1e-15)
        assert f([1, 2, 3], [4]) == (0.0, 0.0, 0.0)
        assert f([5], [5]) == (1.0, 1.0, 1.0)
        # precision is over the items actually returned (k = recItems.length), not topk
        assert f([5, 6], [6, 5, 1, 2])[0] == 1.0
    assert ev.compute_metrics([], [1]) == (0.0, 0.0, 0.0)


def test_metrics_host_mirror_equals_restatement():
    rng = np.random.default_rng(0)
    for _ in range(200):
        k = int(rng.integers(1, 30))
        rec = rng.permutation(60)[:k]
        lab = rng.permutation(60)[:int(rng.integers(1, 12))]
        assert ev.compute_metrics(rec, lab) == pytest.approx(eo.compute_metrics(rec, lab), rel=1e-14, abs=0)


def test_bce_and_evalresult():
    x = np.array([0.0, 2.0, -3.0], np.float32)
    z = np.array([1.0, 0.0, 1.0], np.float32)
    exp = (math.log(2) + (2 + math.log1p(math.exp(-2))) + math.log1p(math.exp(-3)) - (0 * 1 + 2 * 0 + -3 * 1)) / 3
    assert eo.bce_with_logits(x, z) == pytest.approx(exp, rel=1e-6)
    assert ev.bce_with_logits(x, z) == pytest.approx(exp, rel=1e-6)
    r = ev.EvalResult(loss=2.0, count=2)
    r.add_metrics((0.5, 0.25, 1.0))
    r = r + ev.EvalResult(loss=1.0, precision=0.5, recall=0.75, ndcg=0.0, count=2)
    assert str(r) == "{eval loss: 0.7500, precision: 0.250000, recall: 0.250000, ndcg: 0.250000}"
    assert str(r) == str(eo.EvalResult(3.0, 1.0, 1.0, 1.0, 4))


@pytest.mark.gpu
def test_evaluate_vs_oracle(fixture_tree, fixture_w32, oracle, oracle_tree):
    from dismember_amd import Engine
    t = fixture_tree
    E, L, depth = 16, 10, int(t["max_level"])
    ni = (1 << (depth + 1)) - 1
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], depth)
    eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(fixture_w32, E, ni)
    din = oracle.Din(fixture_w32, E, L, ni)
    rng = np.random.default_rng(11)
    N = 40
    items = t["leaf_ids"]
    seqs = rng.choice(items, size=(N, L)).astype(np.int32)
    seqs[rng.random((N, L)) < 0.2] = 0
    labels = [rng.choice(items, size=int(rng.integers(1, 8)), replace=False).astype(np.int32) for _ in range(N)]
    users = rng.integers(0, 12, size=N)
    consumed = {u: rng.choice(items, size=int(rng.integers(5, 60)), replace=False).astype(np.int32) for u in range(12)}
    neg = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], np.int32)
    res, batches = ev.evaluate(eng, seqs, labels, users, consumed, neg, topk=10, candidate_num=40, batch_size=900,
                               return_batches=True)
    ores = eo.evaluate(oracle_tree, din, seqs, labels, users, consumed, batches, topk=10, candidate_num=40)
    assert res.count == ores.count == N
    assert res.loss == pytest.approx(ores.loss, rel=1e-4)
    # metrics are sums of per-user ratios of small integers: per user they are EQUAL whenever the two id lists are equal, so the
    # totals may differ only by the contributions of the users whose lists differ (a near-tie at a cut can flip one)
    ids, _, cnt = eng.tdm_beam_search(seqs, 40, 10, consumed=[consumed[int(u)] for u in users], widen_consumed=True)
    diff_users, bound = 0, np.zeros(3)
    tot_g, tot_o = np.zeros(3), np.zeros(3)
    for i in range(N):
        orec = oracle_tree.recommend_items(din, seqs[i], 10, 40, consumed=consumed[int(users[i])])
        mg = np.array(ev.compute_metrics(ids[i, :cnt[i]], labels[i]))
        mo = np.array(eo.compute_metrics(orec, labels[i]))
        tot_g += mg; tot_o += mo
        if ids[i, :cnt[i]].tolist() == orec.tolist():
            assert (mg == mo).all(), i
        else:
            diff_users += 1
            bound += np.abs(mg - mo)
    assert diff_users <= 2
    got = np.array([res.precision, res.recall, res.ndcg]); want = np.array([ores.precision, ores.recall, ores.ndcg])
    assert np.allclose(got, tot_g, rtol=1e-12, atol=0) and np.allclose(want, tot_o, rtol=1e-12, atol=0)     # the evaluator sums exactly these
    assert (np.abs(got - want) <= bound + 1e-12).all(), (got, want, bound)
    if diff_users == 0:
        assert (res.precision, res.recall, res.ndcg) == pytest.approx((ores.precision, ores.recall, ores.ndcg), rel=1e-12)


def test_otm_dr_eval_helpers():
    # getAllNodes: leaves 7..10 of a 4-leaf mapping (leafLevel 2) -> the leaves and two levels of ancestors
    assert ev.all_nodes([7, 8, 9, 10]) == eo.all_nodes([7, 8, 9, 10]) == {7, 8, 9, 10, 3, 4, 1}
    assert ev.all_nodes([3, 4, 5]) == eo.all_nodes([3, 4, 5]) == {3, 4, 5, 1, 2, 0}
    x, z = np.array([0.0, 2.0, -3.0]), np.array([1.0, 0.0, 1.0])
    exp = math.log(2) + (2 + math.log1p(math.exp(-2))) + math.log1p(math.exp(-3)) - (-3.0)
    assert ev.bce_with_logits_sum(x, z) == pytest.approx(exp, rel=1e-14)
    assert ev.bce_with_logits_sum([], []) == 0.0
    r = ev.OtmEvalResult(1.0, 0.5, 0.25) + ev.OtmEvalResult(1.0, 0.5, 0.25)
    assert str(r / 4) == "{precision: 0.500000, recall: 0.250000, ndcg: 0.125000}"
    d = ev.DrEvalResult([1.0, 2.5], 3.0, 1.0, 2.0, 3.0, 4) + ev.DrEvalResult([1.0, 0.5], 1.0, 1.0, 0.0, 1.0, 4)
    assert str(d) == "eval layer loss: [1, 1.5], rerank loss: 2.0000\n\t\tprecision: 0.250000, recall: 0.250000, ndcg: 0.500000"
    m = d.mean_metrics()
    assert (m.layer_loss, m.rerank_loss, m.precision, m.size, m.count) == ([1.0, 1.5], 2.0, 0.25, 8, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("thread_num", [1, 3])
def test_evaluate_otm_vs_oracle(oracle, thread_num):
    """Evaluator.evaluate of the OTM module (otm/.../evaluation/Evaluator.scala:29-84) on an fp64 model: the device evaluator's
    per-sample lists are the fp64 oracle's (the search is node-for-node equal), so metrics are equal and the loss sum agrees to
    the scores' 1e-9."""
    from dismember_amd import Engine
    from helpers import random_din_weights
    rng = np.random.default_rng(31)
    E, L, leaf_level, beam, topk, N = 32, 10, 8, 12, 7, 37
    NI = (1 << (leaf_level + 1)) - 1
    w = random_din_weights(rng, E, NI, dtype=np.float64, std=0.2, bias_std=0.1)
    eng = Engine(0)
    eng.load_weights_din(w, E, NI)
    odin = oracle.Din(w, E, L, NI)
    first = (1 << leaf_level) - 1
    leaves = first + rng.permutation(1 << leaf_level)[:200]                   # 200 mapped items of the 256 leaves
    allowed = ev.all_nodes(leaves)
    assert allowed == eo.all_nodes(leaves)
    seqs = rng.choice(leaves, size=(N, L)).astype(np.int32)
    seqs[rng.random((N, L)) < 0.2] = -1
    labels = [rng.choice(leaves, size=int(rng.integers(1, 9)), replace=False).tolist() for _ in range(N)]
    users = rng.integers(0, 9, size=N)
    consumed = {u: rng.choice(leaves, size=int(rng.integers(3, 40)), replace=False) for u in range(9)}
    loss, res = ev.evaluate_otm(eng, seqs, labels, users, consumed, allowed, leaf_level, topk, total_eval_batch_size=240,
                                beam_size=beam, thread_num=thread_num)
    oloss, ores = eo.evaluate_otm(lambda s: oracle.otm_beam_search(odin, s, leaf_level, beam), seqs, labels, users, consumed, allowed,
                                  topk, 240, beam, thread_num=thread_num)
    assert (res.precision, res.recall, res.ndcg) == pytest.approx(ores, rel=1e-12, abs=0)
    assert res.recall > 0
    assert loss == pytest.approx(oloss, rel=1e-8)
    eng.close()


@pytest.mark.gpu
def test_evaluate_dr_vs_oracle():
    """The metrics fold of the Deep-Retrieval evaluator (deep-retrieval/.../evaluation/Evaluator.scala:41-70, recommendItems
    :108-129) on an fp64 model: consumed items dropped BEFORE the re-rank cut — the device list asked for topk + |consumed|."""
    from test_gpu_dr import make, histories
    K, D, L, E, num_item, beam, topk, N = 5, 2, 6, 16, 400, 6, 10, 33      # 25 paths, 2 per item: every beam reaches items
    eng, orc, w, rng = make(K, D, L, E, num_item, 5, np.float64)
    seqs = histories(rng, N, L, num_item)
    users = rng.integers(0, 7, size=N)
    consumed = {u: rng.choice(num_item, size=int(rng.integers(1, 120)), replace=False) for u in range(7)}
    labels = [rng.choice(num_item, size=int(rng.integers(1, 12)), replace=False).tolist() for _ in range(N)]
    ids0, _, cnt0 = eng.dr_recommend(seqs, beam, 40)
    for u in range(0, N, 2):                                      # every other sample: labels the search can find (some consumed, some not)
        if cnt0[u] > 3:
            labels[u] = ids0[u, [0, 2, cnt0[u] - 1]].tolist() + labels[u][:2]
    res = ev.evaluate_dr(eng, seqs, labels, users, consumed, topk, beam, batch_size=10, num_layer=D)
    op, or_, og = eo.evaluate_dr_metrics(orc, seqs, labels, users, consumed, topk, beam)
    assert res.size == N and res.count == 4 and res.layer_loss == [0.0] * D
    assert (res.precision, res.recall, res.ndcg) == pytest.approx((op, or_, og), rel=1e-12, abs=0)
    assert res.recall > 0
    eng.close()


@pytest.mark.gpu
def test_evaluate_otm_on_the_reference_model_and_mapping(oracle, fixture_w64):
    """The device evaluator on the reference's own artefacts: bundled trained DIN[Double], bundled item -> node mapping, the bundled
    interactions split by LocalDataSet.generateSamples — per-sample lists equal the fp64 oracle's, so the metrics are equal."""
    from dismember_amd import Engine, otm_data as od, tasks
    m = np.load(os.path.join(os.path.dirname(__file__), "golden", "otm_mapping.npy"))
    mapping = {int(a): int(b) for a, b in m}
    s = tasks._otm_sample(os.path.join(os.path.dirname(__file__), "golden", "example_data.npz"))
    E, L, leaf_level, beam, topk = 16, 10, 12, 20, 10
    NI = (1 << (leaf_level + 1)) - 1
    consumed, _, evals = od.generate_samples(s, mapping, L, 2, 0.8, 5)
    evals = evals[:600]
    seqs = np.array([e[0] for e in evals], np.int32)
    labels, users = [e[1] for e in evals], np.array([e[2] for e in evals])
    allowed = ev.all_nodes(list(mapping.values()))
    eng = Engine(0)
    eng.load_weights_din(fixture_w64, E, NI)
    loss, res = ev.evaluate_otm(eng, seqs, labels, users, consumed, allowed, leaf_level, topk, 8192, beam)
    odin = oracle.Din(fixture_w64, E, L, NI)
    oloss, ores = eo.evaluate_otm(lambda sq: oracle.otm_beam_search(odin, np.asarray(sq, np.int32), leaf_level, beam), seqs, labels, users, consumed,
                                  allowed, topk, 8192, beam)
    assert (res.precision, res.recall, res.ndcg) == pytest.approx(ores, rel=1e-12, abs=0) and res.recall > 0.005
    assert loss == pytest.approx(oloss, rel=1e-8)
    eng.close()
