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
