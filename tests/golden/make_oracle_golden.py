#!/usr/bin/env python3
"""Pins the oracle: writes oracle_outputs.json = the restatement's outputs on the reference's
bundled model/tree fixtures ("restatement-derived, not JVM-derived" — the reference cannot run
in the build container).  A later change to oracle/ that alters any of these
values fails tests/test_oracle.py::test_golden_outputs_pinned."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402

QUERIES = [
    [0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882],   # examples/.../tdm/package.scala:115
    [1, 2, 3, 4, 5, 6, 7, 89, 2628, 1681],                  # DeepRetrievalSpec.scala:110
    [0] * 10,                                               # all padding
    [0, 0, 0, 0, 0, 0, 0, 0, 0, 2126],
    [3952, 3951, 1, 2, 3, 4, 5, 6, 7, 8],
]


def main():
    po.build(force=True)
    t = np.load(os.path.join(HERE, "tdm_tree.npz"))
    tree = po.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    din = po.Din(np.load(os.path.join(HERE, "din_f32.npy")), 16, 10, 8191)
    out = {"tdm": [], "otm": [], "din_f32": {}, "din_f64": {}}
    for q in QUERIES:
        for topk, beam in ((10, 20), (3, 20), (200, 200), (7, 5)):
            ids, sc = tree.recommend(din, q, topk, beam)
            out["tdm"].append(dict(query=q, topk=topk, beam=beam, ids=ids.tolist(),
                                   logits=[float(np.float32(x)) for x in sc]))
    rng = np.random.default_rng(20250523)
    codes = rng.integers(0, 8191, 64).astype(np.int32)
    seqs = rng.integers(0, 8191, (64, 10)).astype(np.int32)
    seqs[rng.random((64, 10)) < 0.2] = -1
    pad = np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32)
    out["din_f32"] = dict(seed=20250523, logits=[float(x) for x in din.forward(codes, seqs, pad)])
    din64 = po.Din(np.load(os.path.join(HERE, "din_f64.npy")), 16, 10, 8191)
    out["din_f64"] = dict(seed=20250523, logits=[float(x) for x in din64.forward(codes, seqs, pad)])
    m = np.load(os.path.join(HERE, "otm_mapping.npy"))
    item2node = {int(a): int(b) for a, b in m}
    for q in QUERIES:
        sc_codes = [item2node.get(i, -1) for i in q]
        ids, sc = po.otm_beam_search(din64, sc_codes, 12, 20)
        out["otm"].append(dict(query=q, beam=20, leaf_level=12, node_ids=ids.tolist(), scores=[float(x) for x in sc]))
    with open(os.path.join(HERE, "oracle_outputs.json"), "w") as f:
        json.dump(out, f)
    print("wrote oracle_outputs.json")


if __name__ == "__main__":
    main()
