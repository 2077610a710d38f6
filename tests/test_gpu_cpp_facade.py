"""The C++ host facade (include/dismember.hpp: dm::TDM / OTM / JTM / Metrics over the C ABI) on the GPU: a small C++
program (tests/cpp/facade_test.cpp, built here with g++) against the Python mirror and the CPU oracle on the reference's
bundled tree + trained models."""
import json
import re
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade_matches_python_mirror_and_oracle(tmp_path, fixture_tree, fixture_w32, fixture_w64, fixture_otm_mapping,
                                                      oracle, oracle_tree, oracle_din32):
    from dismember_amd import Engine, TDM, OTM
    from dismember_amd.jtm import JTM
    exe = str(tmp_path / "facade_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "dismember_amd"), "-ldismember_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "dismember_amd"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    t = fixture_tree
    d = tmp_path
    L, topk, beam, E = 10, 10, 20, 16
    rng = np.random.default_rng(8)
    query = np.array([0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882], np.int32)
    consumed = rng.choice(t["leaf_ids"], 37, replace=False).astype(np.int32)
    batch = rng.choice(t["leaf_ids"], size=(9, L)).astype(np.int32)
    n_rows = 600
    row_items = rng.choice(t["leaf_ids"], n_rows).astype(np.int32)
    rows = rng.choice(t["leaf_ids"], size=(n_rows, L)).astype(np.int32)
    omap = fixture_otm_mapping.astype(np.int32)                      # [n, 2] item, node
    otm_query = omap[rng.choice(len(omap), L), 0].astype(np.int32)
    np.array([int(t["max_level"]), E, topk, beam, L], np.int32).tofile(d / "meta.i32")
    for name, arr, dt in [("codes.i32", t["codes"], np.int32), ("ids.i32", t["ids"], np.int32), ("is_leaf.u8", t["is_leaf"], np.uint8),
                          ("leaf_ids.i32", t["leaf_ids"], np.int32), ("leaf_codes.i32", t["leaf_codes"], np.int32),
                          ("w32.f32", fixture_w32, np.float32), ("w64.f64", fixture_w64, np.float64), ("query.i32", query, np.int32),
                          ("consumed.i32", consumed, np.int32), ("batch.i32", batch, np.int32), ("jtm_row_items.i32", row_items, np.int32),
                          ("jtm_rows.i32", rows, np.int32), ("otm_mapping.i32", omap, np.int32), ("otm_query.i32", otm_query, np.int32)]:
        np.ascontiguousarray(arr, dtype=dt).tofile(d / name)
    # Deep-Retrieval inputs (fp64 model, like the reference)
    from dismember_amd import synth
    dE, dL, dK, dD, dn, dbeam, dtopk = 16, 6, 10, 3, 300, 30, 8
    drng = np.random.default_rng(12)
    dw = synth.make_dr_model(dn, dK, dD, dL, dE, drng)
    dpi = synth.dr_path_items(synth.make_dr_paths(dn, dK, dD, 2, drng))
    np.array([dE, dL, dK, dD, dn, dbeam, dtopk], np.int32).tofile(d / "dr_meta.i32")
    dw["layer_emb"].tofile(d / "dr_layer_emb.f64")
    for i in range(dD):
        dw["layer_w"][i].tofile(d / ("dr_w%d.f64" % i)); dw["layer_b"][i].tofile(d / ("dr_b%d.f64" % i))
    for k in ("rerank_emb", "rerank_w", "rerank_b", "softmax_w", "softmax_b"):
        dw[k].tofile(d / ("dr_%s.f64" % k))
    dpi[0].astype(np.int32).tofile(d / "dr_path_nodes.i32"); dpi[1].astype(np.int64).tofile(d / "dr_item_off.i64")
    dpi[2].astype(np.int32).tofile(d / "dr_items.i32")
    dr_query_ids = drng.integers(0, dn, dL)
    (1000 + 3 * dr_query_ids).astype(np.int32).tofile(d / "dr_query.i32")
    raw = subprocess.check_output([exe, str(d)], env=dict(os.environ), timeout=300).decode()
    # RCCL prints its version banner on stdout when the program creates its one-rank communicator: keep the program's own lines
    out = json.loads("\n".join(ln for ln in raw.splitlines() if not re.match(r"^(RCCL version|HIP version|ROCm version|Hostname|Librccl path)\s*:", ln)))

    # ---- the Python mirror over the same library
    eng = Engine(0)
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"]))
    eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(fixture_w32, E, 8191)
    tdm = TDM(eng, "din")
    py = tdm.recommend(query, topk, beam)
    assert [r[0] for r in out["tdm_recommend"]] == [r[0] for r in py]
    assert out["tdm_recommend_clone"] == out["tdm_recommend"]            # dm::Engine::cloneEngine(): the worker's engine reads the owner's model
    assert np.allclose([r[1] for r in out["tdm_recommend"]], [r[1] for r in py], rtol=0, atol=1e-15)    # libm vs numpy exp: 1 ulp
    assert out["items_plain"] == tdm.recommend_items(query, topk, beam).tolist()
    assert out["items_consumed"] == tdm.recommend_items(query, topk, beam, consumed_items=consumed).tolist()
    assert not (set(out["items_consumed"]) & set(consumed.tolist()))
    assert out["batch_first_ids"] == [r[0][0] for r in tdm.recommend(batch, topk, beam)]
    # ---- and the CPU oracle (ids; the canonical query is pinned in tests/golden/oracle_outputs.json as well)
    oi, _ = oracle_tree.recommend(oracle_din32, query, topk, beam)
    assert [r[0] for r in out["tdm_recommend"]] == oi.tolist()
    # ---- metrics known answer (tests/test_evaluation.py)
    assert out["metrics"] == pytest.approx([0.5, 2 / 3.0, 1.5 / (1 + np.log(2) / np.log(3))], rel=1e-15)
    # ---- JTM.optimize: C++ driver == Python driver (same entry points, same item order)
    item_rows = {}
    for it, r in zip(row_items.tolist(), rows):
        item_rows.setdefault(it, []).append(r)
    item_rows = {k: np.concatenate(v) for k, v in item_rows.items()}
    proj = JTM(eng, t["leaf_ids"], t["leaf_codes"], 12, item_rows, gap=2, seq_len=L).optimize()
    assert {a: b for a, b in out["jtm_projection"]} == proj
    assert out["jtm_all_equal"] is True              # dm::JTM::optimizeAll({&eng}) and optimize() under a one-rank communicator
    codes = np.array([b for _, b in out["jtm_projection"]])
    assert len(codes) == 3706 and codes.min() >= 4095 and codes.max() <= 8190 and np.bincount(codes).max() == 1
    # ---- OTM
    e2 = Engine(0)
    e2.load_weights_din(fixture_w64, E, 8191)
    otm = OTM(e2, {int(a): int(b) for a, b in omap})
    pyo = otm.recommend(otm_query, topk, beam)
    assert [r[0] for r in out["otm_recommend"]] == [r[0] for r in pyo]
    assert np.allclose([r[1] for r in out["otm_recommend"]], [r[1] for r in pyo], rtol=0, atol=1e-15)
    # ---- Deep-Retrieval: the fp64 oracle's recommendation, mapped through the same item <-> id maps
    from oracle import pyoracle as po
    orc = po.DeepRetrieval(dw, dE, dL, dK, dD, dn, path_items=dpi)
    oi, osc = orc.recommend(dr_query_ids.astype(np.int32), dtopk, dbeam)
    assert [r[0] for r in out["dr_recommend"]] == (1000 + 3 * oi).tolist() and len(oi) > 0
    assert np.allclose([r[1] for r in out["dr_recommend"]], 1.0 / (1.0 + np.exp(-osc)), rtol=1e-9)
    assert out["error_code_L40"] == -1                       # DM_ERR_INVALID surfaces as dm::Error
