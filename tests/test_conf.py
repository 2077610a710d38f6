"""`.conf` reader + task table (row g: north_star's ".conf-driven tasks"): Property.readConf / getOrStop semantics
(scalann/src/main/scala/com/mass/scalann/utils/Property.scala:12-71), the key sets of the reference's four config files
(golden fixture tests/golden/conf_keys.json, made by make_conf_fixture.py), and the repo's own configs for BASELINE's five
configurations.  The C++ mirror (dm::Property, include/dismember.hpp) is checked against the Python reader."""
import json
import os
import subprocess

import pytest

from dismember_amd import conf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "conf_keys.json")))
REF = "/root/reference/configs"


def test_read_conf_semantics(tmp_path):
    p = tmp_path / "x.conf"
    p.write_text("# comment\n"
                 "model.embed_size   16\n"
                 "model.seq_len\t 10   \n"
                 " model.leading_blank 1\n"          # does not START with the prefix: dropped (Property.scala:31)
                 "model.three tokens here\n"         # != 2 tokens: dropped (:33)
                 "model.novalue\n"
                 "modelx.odd 5\n"                    # startsWith(prefix) is all the reference checks
                 "init.seq_len 7\n"
                 "model.embed_size 32\n"             # Map(lines: _*): the last occurrence wins
                 "model.path a/b.bin\r\n")
    c = conf.read_conf(str(p), "model")
    assert c == {"embed_size": "32", "seq_len": "10", ".odd": "5", "path": "a/b.bin"}
    assert conf.read_conf(str(p), "model", truncate=False)["model.seq_len"] == "10"
    assert conf.read_conf(str(p), "init") == {"seq_len": "7"}
    assert conf.get_or_stop(c, "seq_len") == "10"
    with pytest.raises(ValueError, match="failed to read parameter: beam_size in conf file"):
        conf.get_or_stop(c, "beam_size")
    with pytest.raises(ValueError, match="doesn't exist"):
        conf.read_conf(str(tmp_path / "missing.conf"), "model")
    assert conf.core_number(0) == (os.cpu_count() or 1) and conf.core_number(-3) == (os.cpu_count() or 1) and conf.core_number(5) == 5


def test_task_table_covers_the_reference_key_sets():
    """Every key a task reads exists in the reference's conf for that prefix, and every required key is there."""
    for task, (flag, prefix, res, table) in conf.TASKS.items():
        keys = set(GOLD[res][prefix]["keys"])
        for key, _, default in table:
            if default is conf.REQ:
                assert key in keys, (task, key)
        assert flag in ("tdmConfFile", "jtmConfFile", "otmConfFile", "drConfFile")
    assert sorted(conf.TASKS) == ["DRCoordinateDescent", "DRTrainDeepModel", "JTMInitializeTree", "JTMTrainDeepModel",
                                  "JTMTreeLearning", "OTMConstructTree", "OTMTrainDeepModel", "TDMClusterTree",
                                  "TDMInitializeTree", "TDMTrainDeepModel"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_reference_confs_load_unchanged():
    for res, prefixes in GOLD.items():
        for prefix, g in prefixes.items():
            c = conf.read_conf(os.path.join(REF, res + ".conf"), prefix)
            assert sorted(c) == g["keys"]
            assert {k: c[k] for k in g["hot_values"]} == g["hot_values"]
    p = conf.task_params("TDMTrainDeepModel", os.path.join(REF, "tdm.conf"))
    assert p["embed_size"] == 16 and p["beam_size"] == 20 and p["topk_number"] == 10 and p["use_mask"] and p["deep_model"] == "din"
    assert p["layer_negative_counts_list"][:4] == [0, 1, 2, 3] and len(p["layer_negative_counts_list"]) == 23
    assert p["sample_with_probability"] is False and p["learning_rate"] == 1e-4 and p["thread_number"] == (os.cpu_count() or 1)
    assert conf.task_params("JTMTreeLearning", os.path.join(REF, "jtm.conf"))["gap"] == 2
    assert conf.task_params("OTMTrainDeepModel", os.path.join(REF, "otm.conf"))["label_num"] == 5
    assert conf.task_params("DRCoordinateDescent", os.path.join(REF, "deep-retrieval.conf"))["penalty_factor"] == 3e-6
    assert conf.task_params("TDMInitializeTree", os.path.join(REF, "tdm.conf"))["user_consumed_path"] == "data/user_consumed.txt"


def test_repo_configs_cover_baseline_configs():
    c1 = os.path.join(ROOT, "configs", "c1_tdm_movielens.conf")
    for prefix in ("init", "model", "cluster"):
        assert set(conf.read_conf(c1, prefix)) >= set(GOLD["tdm"][prefix]["keys"]) - {"cluster_iter"} or prefix == "cluster"
    assert set(conf.read_conf(c1, "model")) == set(GOLD["tdm"]["model"]["keys"])
    p = conf.task_params("TDMTrainDeepModel", c1)
    assert (p["embed_size"], p["beam_size"], p["topk_number"], p["seq_len"]) == (16, 20, 10, 10)
    p2 = conf.task_params("TDMTrainDeepModel", os.path.join(ROOT, "configs", "c2_tdm_serve_1m.conf"))
    assert (p2["embed_size"], p2["beam_size"], p2["topk_number"]) == (128, 200, 200)
    p3 = conf.task_params("OTMTrainDeepModel", os.path.join(ROOT, "configs", "c3_otm_10m.conf"))
    assert p3["beam_size"] == 200 and p3["label_num"] == 5 and p3["train_batch_size"] == 8192
    assert set(conf.read_conf(os.path.join(ROOT, "configs", "c3_otm_10m.conf"), "model")) == set(GOLD["otm"]["model"]["keys"])
    assert conf.task_params("OTMConstructTree", os.path.join(ROOT, "configs", "c3_otm_10m.conf"))["gap"] == 2
    assert conf.task_params("JTMTreeLearning", os.path.join(ROOT, "configs", "c4_jtm_10m.conf"))["gap"] == 2
    p5 = conf.task_params("DRTrainDeepModel", os.path.join(ROOT, "configs", "c5_dr_10m.conf"))
    assert (p5["num_layer"], p5["num_node"], p5["beam_size"], p5["embed_size"]) == (3, 1000, 50, 128)
    assert conf.task_params("DRCoordinateDescent", os.path.join(ROOT, "configs", "c5_dr_10m.conf"))["train_mode"] == "streaming"


def test_cpp_property_matches_python(tmp_path):
    src = tmp_path / "p.cpp"
    src.write_text('#include <cstdio>\n#include "dismember.hpp"\n'
                   'int main(int c, char **v) { auto m = dm::Property::readConf(v[1], v[2]);\n'
                   '  for (auto &kv : m) std::printf("%s=%s\\n", kv.first.c_str(), kv.second.c_str());\n'
                   '  try { dm::Property::getOrStop(m, "no_such_key"); } catch (const std::invalid_argument &e) { std::printf("ERR %s\\n", e.what()); }\n'
                   '  std::printf("cores %d %d\\n", dm::Property::getCoreNumber(0) > 0, dm::Property::getCoreNumber(3)); }\n')
    exe = str(tmp_path / "p")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe, "-pthread"])
    for path, prefix in [(os.path.join(ROOT, "configs", "c1_tdm_movielens.conf"), "model"),
                         (os.path.join(ROOT, "configs", "c5_dr_10m.conf"), "cd")]:
        out = subprocess.check_output([exe, path, prefix]).decode().splitlines()
        got = dict(l.split("=", 1) for l in out if "=" in l and not l.startswith("ERR"))
        assert got == conf.read_conf(path, prefix)
        assert "ERR failed to read parameter: no_such_key in conf file" in out and "cores 1 3" in out
