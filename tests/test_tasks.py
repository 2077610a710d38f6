"""The reference's conf-driven tasks (examples/src/main/scala/com/mass/retrieval/{tdm,jtm}/*.scala) through dismember_amd.tasks:
TDMInitializeTree on the CPU (host integer logic only), TDMTrainDeepModel and JTMTreeLearning on the device."""
import os

import numpy as np
import pytest

from dismember_amd import tasks, tree_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _conf(tmp_path, **over):
    """configs/c1_tdm_movielens.conf with its output paths moved under tmp_path and selected keys overridden."""
    out = []
    for line in open(os.path.join(ROOT, "configs", "c1_tdm_movielens.conf")):
        s = line.strip()
        if s and not s.startswith("#"):
            key, val = s.split(None, 1)
            if val.startswith("gpurun_out/c1/"):
                val = str(tmp_path / val[len("gpurun_out/c1/"):])
            elif val.startswith("tests/"):
                val = os.path.join(ROOT, val)
            if key in over:
                val = str(over.pop(key))
            line = "%s %s\n" % (key, val)
        out.append(line)
    for k, v in over.items():
        out.append("%s %s\n" % (k, v))
    p = tmp_path / "task.conf"
    p.write_text("".join(out))
    return str(p)


def test_tdm_initialize_tree_task(tmp_path):
    conf = _conf(tmp_path)
    r = tasks.tdm_initialize_tree(conf)
    p = r["params"]
    assert r["n_items"] == 3325                                   # the bundled sample's catalogue
    for k in ("train_path", "eval_path", "stat_path", "leaf_id_path", "tree_protobuf_path", "user_consumed_path"):
        assert os.path.getsize(p[k]) > 0, k
    t = tree_io.read_tree_file(p["tree_protobuf_path"])
    assert t["max_level"] == 12 and len(t["leaf_ids"]) == 3325
    assert (t["leaf_codes"] >= (1 << 12) - 1).all()               # TreeInitSpec.scala:44-47: every leaf on the last level
    seqs, tgts = tree_io.read_train_data(open(p["train_path"]))
    assert seqs.shape == (r["n_train"], 10) or seqs.shape[0] <= r["n_train"]
    eseq, elab, euser = tree_io.read_eval_data(open(p["eval_path"]), 10)
    assert len(elab) == r["n_eval"] and eseq.shape[1] == 10
    # the command line: same task through main(); unknown / out-of-scope tasks do not run
    assert tasks.main(["TDMInitializeTree", "--tdmConfFile", conf, "--quiet"]) == 0
    assert tasks.main(["TDMClusterTree", "--tdmConfFile", conf]) == 3
    assert tasks.main(["NoSuchTask"]) == 2 and tasks.main(["TDMInitializeTree"]) == 2


def test_missing_key_stops_like_get_or_stop(tmp_path):
    conf = _conf(tmp_path)
    txt = "".join(l for l in open(conf) if not l.startswith("init.data_path"))
    open(conf, "w").write(txt)
    with pytest.raises(Exception) as e:
        tasks.tdm_initialize_tree(conf)
    assert "data_path" in str(e.value)


@pytest.mark.gpu
def test_tdm_train_and_jtm_tree_learning_tasks(tmp_path):
    """TDMInitializeTree -> TDMTrainDeepModel -> JTMTreeLearning from one conf file: the loss goes down, the evaluator reports at
    the progress interval, the model file reloads into a fresh handle with identical recommendations (TdmModelTrainSpec.scala:85-96),
    and the learned projection is a bijection onto leaf codes (JtmSpec.scala:37-51) written as a loadable tree file."""
    from dismember_amd import Engine, TDM
    conf = _conf(tmp_path, **{"model.iteration_number": 120, "model.show_progress_interval": 60,
                              "tree.deep_model": "DIN", "tree.data_path": str(tmp_path / "train_data.csv"),
                              "tree.tree_protobuf_path": str(tmp_path / "tdm_tree.bin"), "tree.model_path": str(tmp_path / "tdm_model.bin"),
                              "tree.gap": 2, "tree.seq_len": 10, "tree.hierarchical_preference": "false", "tree.min_level": 0,
                              "tree.thread_number": 0})
    tasks.tdm_initialize_tree(conf)
    r = tasks.tdm_train_deep_model(conf, time_recommend=False)
    losses = r["losses"]
    assert len(losses) == 120 and np.mean(losses[-15:]) < 0.9 * np.mean(losses[:15])
    assert [it for it, _ in r["eval"]] == [60, 120]
    assert r["eval"][-1][1]["loss"] < 0.75 and len(r["recommendation"]) == 3
    eng = r["engine"]
    q = np.array([0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882], np.int32)
    before = TDM(eng).recommend(q, 10, 20)
    e2 = Engine(0)
    after = TDM.load_model(e2, r["params"]["model_path"]).recommend(q, 10, 20)
    assert before == after
    e2.close()
    eng.close()
    old = tree_io.read_tree_file(str(tmp_path / "tdm_tree.bin"))
    j = tasks.jtm_tree_learning(conf)
    proj = j["projection"]
    new = tree_io.read_tree_file(str(tmp_path / "tdm_tree.bin"))
    assert sorted(proj) == sorted(old["leaf_ids"].tolist())
    codes = np.array(list(proj.values()))
    assert len(set(codes.tolist())) == len(codes) and (codes >= (1 << 12) - 1).all() and (codes < (1 << 13) - 1).all()
    assert dict(zip(new["leaf_ids"].tolist(), new["leaf_codes"].tolist())) == {int(k): int(v) for k, v in proj.items()}
    assert new["max_level"] == 12
    j["engine"].load_tree_file(str(tmp_path / "tdm_tree.bin"))          # the written file is a valid index
    j["engine"].close()


def _otm_conf(tmp_path, **over):
    out = []
    for line in open(os.path.join(ROOT, "configs", "c1_otm_movielens.conf")):
        s = line.strip()
        if s and not s.startswith("#"):
            key, val = s.split(None, 1)
            if val.startswith("gpurun_out/c1/"):
                val = str(tmp_path / val[len("gpurun_out/c1/"):])
            elif val.startswith("tests/"):
                val = os.path.join(ROOT, val)
            if key in over:
                val = str(over.pop(key))
            line = "%s %s\n" % (key, val)
        out.append(line)
    p = tmp_path / "otm.conf"
    p.write_text("".join(out))
    return str(p)


def test_otm_dataset_and_mapping_files(tmp_path):
    """LocalDataSet of the OTM module on the bundled sample (host logic): the initial mapping is a bijection onto leaves of level
    upperLog2(#items) (TreeConstructionSpec.scala:38-48), every training window has seq_len history slots and <= label_num labels, an
    evaluation sample's labels follow its history, and the mapping file round-trips."""
    from dismember_amd import otm_data as od
    s = tasks._otm_sample(os.path.join(ROOT, "tests", "golden", "example_data.npz"))
    for mode in ("random", "category"):
        m = od.initialize_mapping(s, mode, np.random.default_rng(3))
        nodes = np.array(sorted(m.values()))
        assert len(m) == 3325 and np.unique(nodes).size == 3325 and nodes[0] >= 4095 and nodes[-1] <= 8190
    consumed, train, evals = od.generate_samples(s, m, 10, 2, 0.8, 5)
    assert len(train) > 10 * len(evals) > 0
    assert all(len(t[0]) == 10 and 1 <= len(t[1]) <= 5 for t in train) and all(len(e[0]) == 10 and len(e[1]) >= 1 for e in evals)
    assert all(set(e[1]).isdisjoint(consumed[e[2]]) for e in evals[:200])          # labels lie past the consumed prefix
    seqs = od.item_sequences(s, m, 5, 2, 10, 0.8)
    assert all(v.size % 10 == 0 for v in seqs.values()) and set(seqs) <= set(m)
    od.save_mapping(str(tmp_path / "m.txt"), m)
    assert od.load_mapping(str(tmp_path / "m.txt")) == m
    assert od._sliding([1, 2, 3], 5) == [[1, 2, 3]] and od._sliding([1, 2, 3, 4], 3) == [[1, 2, 3], [2, 3, 4]]


@pytest.mark.gpu
def test_otm_train_and_construct_tree_tasks(tmp_path):
    """OTMTrainDeepModel -> OTMConstructTree from one conf file (the reference's otm.conf values, two epochs): per-level losses fall,
    the evaluator runs at the end of every epoch, the saved model + mapping serve the same recommendations from a fresh handle
    (OtmModelTrainSpec.scala:47-58), and tree construction returns a bijection of the items onto leaves (TreeConstructionSpec.scala:38-48)."""
    from dismember_amd import Engine, OTM
    from dismember_amd import otm_data as od
    conf = _otm_conf(tmp_path, **{"model.epoch_num": 2})
    r = tasks.otm_train_deep_model(conf, time_recommend=False)
    levels = r["epoch_losses"]["epoch 1"]
    assert len(levels) == 12 - 4 and len(levels[0]) == 7                         # leaf level 12, start level floor(log2 20) = 4; 55 869 windows / 8 192
    first, last = np.mean([lv[0] for lv in levels]), np.mean([lv[-1] for lv in r["epoch_losses"]["epoch 2"]])
    assert last < 0.8 * first, (first, last)
    assert [(e, i) for e, i, _, _ in r["eval"]] == [(1, 7), (2, 7)] and r["eval"][-1][2] < r["eval"][0][2] * 1.05
    assert len(r["recommendation"]) == 3
    eng = r["engine"]
    q = [0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882]
    before = OTM(eng, r["mapping"]).recommend(q, 10, 20)
    e2 = Engine(0)
    e2.load_model(r["params"]["model_path"])
    after = OTM(e2, od.load_mapping(r["params"]["mapping_path"])).recommend(q, 10, 20)
    assert before == after
    e2.close(); eng.close()
    t = tasks.otm_construct_tree(conf)
    new = t["mapping"]
    assert set(new) == set(t["old_mapping"]) and len(set(new.values())) == len(new)
    assert min(new.values()) >= 4095 and max(new.values()) <= 8190
    assert od.load_mapping(t["params"]["mapping_path"]) == new
    t["engine"].close()
    assert tasks.main(["OTMConstructTree", "--otmConfFile", conf, "--quiet"]) == 0
