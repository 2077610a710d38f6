"""BASELINE config 1 end to end on the device: the reference's bundled MovieLens sample ->
tree initialisation -> tree file -> device index -> data-parallel-style training steps -> evaluator.
Mirrors tdm/src/test/scala/TdmModelTrainSpec.scala (losses go down; >= 3 recommendations) with configs/tdm.conf's
sizes (embed 16, seq_len 10, min_seq_len 2, split 0.8, topk 10, beam 20, batch 8192 rows)."""
import os

import numpy as np
import pytest

from dismember_amd import evaluation as ev
from dismember_amd import tree_io
from helpers import random_din_weights

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_tree_init_train_evaluate_recommend():
    from dismember_amd import Engine, TDM
    from dismember_amd.trainer import TDMTrainer
    d = np.load(os.path.join(GOLD, "example_data.npz"))
    sample = dict(user=d["user"].astype(int).tolist(), item=d["item"].astype(int).tolist(),
                  timestamp=list(range(len(d["user"]))))
    split = tree_io.split_samples(tree_io.user_sequences(sample), 10, 2, True, 0.8)
    ids, codes, _ = tree_io.gen_codes(d["uniq_item"].astype(int), d["uniq_cat"].astype(int))
    t = tree_io.read_tree_bytes(tree_io.build_tree_bytes(ids, codes, split["stat"]))
    assert t["max_level"] == 12 and len(t["leaf_ids"]) == 3325
    E, L, depth = 16, 10, t["max_level"]
    ni = (1 << (depth + 1)) - 1
    rng = np.random.default_rng(7)
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], depth)
    eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(random_din_weights(rng, E, ni), E, ni)
    neg = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], np.int32)          # configs/tdm.conf, first 13 levels
    seqs = np.array([s for _, s in split["train"]], np.int32)
    train_seq, train_tgt = seqs[:, :L], seqs[:, L]
    eseq = np.array([s for _, s, _ in split["eval"]], np.int32)[:600]
    elab = [np.array(l, np.int32) for _, _, l in split["eval"]][:600]
    euser = np.array([int(n[5:]) for n, _, _ in split["eval"]])[:600]
    consumed = {u: np.array(v, np.int32) for u, v in split["user_consumed"].items()}
    before = ev.evaluate(eng, eseq, elab, euser, consumed, neg, topk=10, candidate_num=20, seed=5).means()
    tr = TDMTrainer(eng, neg, lr=3e-3, seed=1)
    per = int(sum(1 + neg[l] for l in range(1, depth + 1)))
    T = max(1, 8192 // per)
    losses = []
    order = rng.permutation(len(train_tgt))
    for it in range(400):
        idx = order[(it * T) % (len(order) - T):][:T]
        losses.append(tr.step(train_seq[idx], train_tgt[idx]))
    after = ev.evaluate(eng, eseq, elab, euser, consumed, neg, topk=10, candidate_num=20, seed=5).means()
    print("loss %.4f -> %.4f ; eval before %s after %s" % (np.mean(losses[:20]), np.mean(losses[-20:]), before, after))
    assert np.mean(losses[-20:]) < 0.8 * np.mean(losses[:20])
    assert after["loss"] < before["loss"]
    assert after["recall"] > max(2 * before["recall"], 0.01)
    recs = TDM(eng).recommend(eseq[0], 10, 20)
    assert len(recs) >= 3 and all(0.0 <= p <= 1.0 for _, p in recs)          # TdmModelTrainSpec: at least 3 recommendations
