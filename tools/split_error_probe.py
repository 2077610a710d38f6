"""How far each fp32 evaluation of the scorer is from the exact (fp64) value of the same fp32 weights: the CPU oracle
(reference summation order, fp32), the beam kernel with the fp32-input MFMA, the beam kernel with the split-fp16 scorer.
Scores come from the level traces of real searches (every scored (node, user) row).  GPU; python tools/split_error_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_din_weights, random_histories, synthetic_tree   # noqa: E402
from test_gpu_parity import make_engine                                      # noqa: E402
from oracle import pyoracle as po                                            # noqa: E402

for E, depth, n_items, beam, gain in [(128, 12, 4000, 100, 1.0), (128, 12, 4000, 100, 8.0), (64, 11, 2000, 50, 1.0), (32, 10, 1000, 50, 1.0)]:
    rng = np.random.default_rng(E + depth)
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    w[:NI * E] *= np.float32(gain)
    otree = po.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    o32 = po.Din(w, E, 10, NI)
    o64 = po.Din(w.astype(np.float64), E, 10, NI)
    eng = make_engine(t, w, E)
    seqs = random_histories(rng, t["leaf_ids"], 48, 10)
    res = {}
    for mode in ("f32", "split_f16"):
        eng.set_scorer_mode(mode)
        ids, sc, cnt, tc, ts, tn = eng.tdm_beam_search_trace(seqs, beam, beam)
        err, err_o, mag = [], [], []
        for u in range(seqs.shape[0]):
            seq_codes, mask = otree.id_to_code(seqs[u])
            for it in range(tn.shape[1]):
                n = int(tn[u, it])
                if n == 0:
                    continue
                codes = tc[u, it, :n]
                pad = (mask[None, :] + (np.arange(n) * seq_codes.size)[:, None]).reshape(-1)
                rep = np.tile(seq_codes, (n, 1))
                exact = o64.forward(codes, rep, pad)
                err.append(np.abs(ts[u, it, :n].astype(np.float64) - exact))
                err_o.append(np.abs(o32.forward(codes, rep, pad).astype(np.float64) - exact))
                mag.append(np.abs(exact))
        err, err_o, mag = np.concatenate(err), np.concatenate(err_o), np.concatenate(mag)
        res[mode] = (err, err_o, mag)
    m = res["f32"][2]
    print("E=%d depth=%d beam=%d table gain %.0f: %d scored rows, median |logit| %.3g" % (E, depth, beam, gain, m.size, np.median(m)))
    for name, e in (("CPU oracle (fp32, reference order)", res["f32"][1]), ("beam kernel, fp32-input MFMA", res["f32"][0]),
                    ("beam kernel, split-fp16 scorer", res["split_f16"][0])):
        print("   %-36s max abs err %.3g   rms %.3g   max err / median |logit| %.3g" % (name, e.max(), np.sqrt((e ** 2).mean()), e.max() / np.median(m)))
    eng.close()
