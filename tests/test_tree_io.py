"""Tree file writer / reader and tree initialisation: byte-identical against the reference's
bundled tree file, known answers for the integer logic."""
import hashlib
import json
import os

import numpy as np
import pytest

from dismember_amd import tree_io

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_writer_reproduces_bundled_tree_file_bit_for_bit(fixture_tree):
    """TreeBuilder.build on (leaf ids, leaf codes, target statistics) == data/jtm/example_tree.bin, byte for byte
    (size and sha256 of the original recorded by tests/golden/make_fixtures.py)."""
    t = fixture_tree
    meta = json.load(open(os.path.join(GOLD, "tdm_tree_file.json")))
    leaf = t["is_leaf"] == 1
    ids, codes = t["ids"][leaf], t["codes"][leaf]
    stat = dict(zip(t["stat_ids"].tolist(), t["stat_counts"].tolist()))
    perm = np.random.default_rng(0).permutation(len(ids))        # build() sorts by code itself
    b = tree_io.build_tree_bytes(ids[perm], codes[perm], stat)
    assert len(b) == meta["bytes"]
    assert hashlib.sha256(b).hexdigest() == meta["sha256"]
    # and the reader inverts it
    r = tree_io.read_tree_bytes(b)
    for k in ("codes", "ids", "is_leaf", "leaf_ids", "leaf_codes"):
        assert np.array_equal(r[k], t[k]), k
    assert np.array_equal(r["probs"], t["probs"]) and r["max_level"] == int(t["max_level"]) == 12


def test_writer_without_stat_and_flatten():
    # 3 items, codes from gen_codes [2, 4, 3]: max code 4 -> max_level 2, min leaf code 3; code 2 sinks to 5
    b = tree_io.build_tree_bytes([10, 11, 12], [2, 4, 3])
    r = tree_io.read_tree_bytes(b)
    assert r["max_level"] == 2
    assert r["leaf_codes"].tolist() == [3, 4, 5] and r["leaf_ids"].tolist() == [12, 11, 10]
    assert r["codes"].tolist() == [0, 1, 2, 3, 4, 5]
    assert r["ids"].tolist() == [13, 14, 15, 12, 11, 10]            # ancestors: code + (max id + 1)
    assert (r["probs"] == 1.0).all()
    assert tree_io.flatten_leaves([0, 1, 6], 7) == [7, 7, 13]
    assert tree_io.get_ancestors(12, 3) == [5, 2, 0]


def test_gen_codes_known_answer_and_shape():
    ids, codes, uniq = tree_io.gen_codes([5, 3, 5, 9, 3], [1, 0, 1, 0, 0])
    assert uniq == [5, 3, 9]                     # first appearance (leaf id file order)
    assert ids == [3, 9, 5]                      # (category, id)
    assert codes == [2, 4, 3]                    # genCode: upper half -> 2c+1, lower half -> 2c+2
    rng = np.random.default_rng(1)
    n = 1000
    ids, codes, _ = tree_io.gen_codes(rng.permutation(n), rng.integers(0, 7, n))
    assert len(set(codes)) == n
    depth = [int(np.floor(np.log2(c + 1))) for c in codes]
    assert max(depth) - min(depth) <= 1          # balanced halving
    cs = set(codes)
    assert all(2 * c + 1 not in cs and 2 * c + 2 not in cs for c in codes)   # no code is another's ancestor... child


def test_split_samples_known_answer():
    ui = {7: [1, 2, 3, 4, 5, 6, 7], 8: [1, 2], 9: [4, 5, 6]}
    s = tree_io.split_samples(ui, seq_len=4, min_seq_len=2, split_for_eval=True, split_ratio=0.8)
    # user 7: arr = [0,0,1..7], trainNum = ceil(5 * 0.8) = 4
    assert [x for x in s["train"] if x[0].startswith("user_7_")] == [
        ("user_7_0", [0, 0, 1, 2, 3]), ("user_7_1", [0, 1, 2, 3, 4]), ("user_7_2", [1, 2, 3, 4, 5]), ("user_7_3", [2, 3, 4, 5, 6])]
    assert s["user_consumed"][7] == [1, 2, 3, 4, 5, 6] and s["user_consumed"][8] == [1, 2]
    # user 9: 3 items = minSeqLen + 1 -> consumed = all items, one train window, no eval line
    assert s["user_consumed"][9] == [4, 5, 6] and ("user_9_0", [0, 0, 4, 5, 6]) in s["train"]
    # eval of user 7: split point 4 -> sequence arr[4:8] = [3,4,5,6], labels = rest minus consumed = [7]
    assert s["eval"] == [("user_7", [3, 4, 5, 6], [7])]
    assert s["stat"] == {3: 1, 4: 1, 5: 1, 6: 2}
    t = tree_io.split_samples(ui, 4, 2, split_for_eval=False)
    assert [x[1] for x in t["train"] if x[0].startswith("7_")][-1] == [3, 4, 5, 6, 7] and len(t["train"]) == 5 + 1
    assert t["stat"][7] == 1 and t["eval"] == []


def test_read_interactions_and_sequences():
    lines = ["user,item,label,timestamp,genre", "1,10,5,300,a", "1,11,4,100,b", "2,10,3,50,a", "bad,line", "1,10,1,400,a"]
    s = tree_io.read_interactions(lines)
    assert s["user"] == [1, 1, 2, 1] and s["item"] == [10, 11, 10, 10] and s["category"] == [0, 1, 0, 0]
    assert tree_io.user_sequences(s) == {1: [11, 10], 2: [10]}


@pytest.mark.skipif(not os.path.exists("/root/reference/data/example_data.csv"), reason="needs the reference's bundled csv")
def test_pipeline_on_bundled_csv():
    lines = open("/root/reference/data/example_data.csv").read().splitlines()
    b, ids, codes, split = tree_io.initialize_tree(lines, 10, 2, True, 0.8)      # configs/tdm.conf
    r = tree_io.read_tree_bytes(b)
    assert len(ids) == 3325 and r["max_level"] == 12 and len(r["leaf_ids"]) == 3325
    assert sorted(r["leaf_ids"].tolist()) == sorted(ids)
    assert len(split["train"]) == sum(split["stat"].values())
    root = r["probs"][r["codes"] == 0][0]
    assert root == float(sum(split["stat"].values()))


def test_data_files_round_trip(tmp_path):
    ui = {7: [1, 2, 3, 4, 5, 6, 7], 8: [1, 2], 9: [4, 5, 6]}
    s = tree_io.split_samples(ui, seq_len=4, min_seq_len=2, split_for_eval=True, split_ratio=0.8)
    p = [str(tmp_path / n) for n in ("train.csv", "eval.csv", "stat.txt", "consumed.txt")]
    tree_io.write_split_files(s, *p)
    seqs, tgts = tree_io.read_train_data(open(p[0]).read().splitlines())
    assert seqs.shape == (5, 4) and tgts.tolist() == [3, 4, 5, 6, 6]
    assert seqs[0].tolist() == [0, 0, 1, 2]
    es, el, eu = tree_io.read_eval_data(open(p[1]).read().splitlines(), 4)
    assert es.tolist() == [[3, 4, 5, 6]] and el[0].tolist() == [7] and eu.tolist() == [7]
    uc = tree_io.read_user_consumed(open(p[3]).read().splitlines())
    assert uc[7].tolist() == [1, 2, 3, 4, 5, 6] and uc[8].tolist() == [1, 2]
    assert open(p[2]).read().splitlines()[0] == "3, 1"
    # an all-padding sequence is dropped by the train reader (LocalDataSet.scala:154)
    s2, t2 = tree_io.read_train_data(["u_0,0,0,0,0,5", "u_1,0,0,0,2,6"])
    assert t2.tolist() == [6]


def test_jtm_write_tree(fixture_tree):
    """Writing the identity projection of the bundled tree with JTMTree.writeTree's rules gives back the same nodes
    (ids, leaf flags; probabilities: leaves carried over, ancestors = sums)."""
    t = fixture_tree
    leaf = t["is_leaf"] == 1
    ids, codes, probs = t["ids"][leaf], t["codes"][leaf], t["probs"][leaf]
    order = np.argsort(ids)
    off = int(t["ids"][~leaf].min() - t["codes"][~leaf][np.argmin(t["ids"][~leaf])])
    b = tree_io.build_jtm_tree_bytes(ids[order], codes[order], probs[order], int(t["max_level"]), off)
    r = tree_io.read_tree_bytes(b)
    assert np.array_equal(r["codes"], t["codes"]) and np.array_equal(r["ids"], t["ids"]) and np.array_equal(r["is_leaf"], t["is_leaf"])
    assert np.array_equal(r["probs"][leaf], t["probs"][leaf])
    assert sorted(zip(r["leaf_ids"].tolist(), r["leaf_codes"].tolist())) == sorted(zip(t["leaf_ids"].tolist(), t["leaf_codes"].tolist()))
    # root = total of all leaf probabilities (items without statistics carry 1.0 here, unlike TreeBuilder's 0)
    assert r["probs"][0] == np.float32(probs.sum())
