"""Can the reference's bundled TRAINED DIN weights pin the oracle's parameter layout and forward semantics?  (round-4 verdict, next #5)

The bundled models (data/jtm/example_model.bin -> tests/golden/din_f32.npy, data/otm/example_model.bin -> din_f64.npy) encode the
reference's forward function: read with a wrong layout they should explain the bundled interactions (tests/golden/example_data.npz on
the bundled tree, tests/golden/tdm_tree.npz) worse than read correctly.  This script builds TDM samples the way the reference's
trainer does — sliding 10-item windows per user in time order, the target's ancestors as positives, `layer_negative_counts` = l uniform
negatives at level l (configs/tdm.conf:25, NegativeSampler.scala:76-114) — scores them with the CPU oracle under the loaded vector and
under perturbed restatements, and prints mean BCE (the training objective) and AUC.

What it finds (recorded in DESIGN.md §5): the bundled model is a test artefact, not a converged model — its scores barely depend on the
history, its BCE (0.50) is above the level prior alone (0.37), its AUC is 0.45.  It separates GROSS mis-readings robustly (concat
order [att; item]: +0.03..0.04 BCE on every seed; l1.b <-> l2.W: +0.4; sign of l2.W: +1.4) but not the orientation of l1.W or att.W
(+-0.01, sign depends on the sample).  tests/test_oracle.py::test_trained_weights_pin_layout asserts exactly the robust part.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E, L, NI = 16, 10, 8191


def build_samples(tree_npz, data_npz, po, n_windows, seed):
    t, d = tree_npz, data_npz
    tree = po.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    user, item = d["user"].astype(np.int64), d["item"].astype(np.int64)
    rng = np.random.default_rng(seed)
    order = np.argsort(user, kind="stable")            # interactions are stored in time order: stable sort keeps it per user
    u_s, i_s = user[order], item[order]
    segs = np.split(np.arange(u_s.size), np.flatnonzero(np.diff(u_s)) + 1)
    id2code = dict(zip(t["leaf_ids"].tolist(), t["leaf_codes"].tolist()))
    exists = set(t["codes"].tolist())
    maxl = int(t["max_level"])
    lvl_nodes = {l: np.array([c for c in range((1 << l) - 1, (2 << l) - 1) if c in exists]) for l in range(1, maxl + 1)}
    rows_c, rows_s, rows_y, pads, n = [], [], [], [], 0
    taken = 0
    for seg in segs:
        its = i_s[seg]
        if its.size < 12:
            continue
        k = int(rng.integers(11, its.size))
        codes, mask = tree.id_to_code(its[k - 10:k].astype(np.int32))
        c = id2code[int(its[k])]
        while c > 0:
            l = int(np.floor(np.log2(c + 1)))
            cand = lvl_nodes[l][lvl_nodes[l] != c]
            negs = rng.choice(cand, size=min(l, cand.size), replace=False)
            for node, y in [(c, 1)] + [(int(x), 0) for x in negs]:
                rows_c.append(node); rows_s.append(codes); rows_y.append(y); pads.extend((n * L + mask).tolist()); n += 1
            c = (c - 1) >> 1
        taken += 1
        if taken >= n_windows:
            break
    return (np.array(rows_c, np.int32), np.array(rows_s, np.int32), np.array(rows_y, np.float64), np.array(pads, np.int32))


def perturbations(w):
    off = NI * E
    att = w[off:off + E * E].reshape(E, E)
    l1 = w[off + E * E:off + 3 * E * E].reshape(E, 2 * E)
    tail = off + 3 * E * E
    out = {"as loaded": w}
    v = w.copy(); v[off + E * E:off + 3 * E * E] = l1.reshape(2 * E, E).T.reshape(-1); out["l1.W read as [in][out]"] = v
    v = w.copy(); v[off:off + E * E] = att.T.reshape(-1); out["att.W transposed"] = v
    v = w.copy(); v[off + E * E:off + 3 * E * E] = np.concatenate([l1[:, E:], l1[:, :E]], 1).reshape(-1); out["concat order [att; item]"] = v
    v = w.copy(); v[:off] = w[:off].reshape(E, NI).T.reshape(-1); out["emb read as [E][index]"] = v
    v = w.copy(); v[tail:tail + E] = w[tail + E:tail + 2 * E]; v[tail + E:tail + 2 * E] = w[tail:tail + E]; out["l1.b <-> l2.W"] = v
    v = w.copy(); v[tail + E:tail + 2 * E] = -w[tail + E:tail + 2 * E]; out["l2.W sign flipped"] = v
    return out


def bce(s, y):
    return float(np.mean(np.maximum(s, 0) - s * y + np.log1p(np.exp(-np.abs(s)))))


def auc(s, y):
    o = np.argsort(s, kind="stable")
    r = np.empty(s.size); r[o] = np.arange(1, s.size + 1)
    npos = y.sum(); nneg = y.size - npos
    return float((r[y == 1].sum() - npos * (npos + 1) / 2) / (npos * nneg))


def evaluate(po, w, samples):
    rows_c, rows_s, y, pads = samples
    res = {}
    for name, v in perturbations(w).items():
        s = po.Din(v.copy(), E, L, NI).forward(rows_c, rows_s, pads).astype(np.float64)
        res[name] = (bce(s, y), auc(s, y))
    lv = np.floor(np.log2(rows_c + 1))
    res["(level prior alone: logit = -log l)"] = (bce(-np.log(lv), y), float("nan"))
    return res


if __name__ == "__main__":
    from oracle import pyoracle as po
    po.build()
    g = os.path.join(ROOT, "tests", "golden")
    t, d, w = np.load(os.path.join(g, "tdm_tree.npz")), np.load(os.path.join(g, "example_data.npz")), np.load(os.path.join(g, "din_f32.npy"))
    for seed in (1, 2, 3):
        samples = build_samples(t, d, po, 600, seed)
        print("seed %d: %d rows" % (seed, samples[0].size))
        for name, (b, a) in evaluate(po, w, samples).items():
            print("  %-38s BCE %.4f  AUC %.4f" % (name, b, a))
