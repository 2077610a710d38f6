"""Randomised Deep-Retrieval sweep (GPU, fp64 model): paths bit-exact and probabilities within 1e-9 of the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dismember_amd import Engine, synth
from oracle import pyoracle as po
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(n_cfg):
    E = int(rng.choice([16, 32, 48, 64]))
    K = int(rng.integers(2, 400))
    D = int(rng.integers(2, 5))
    L = int(rng.integers(1, 13))
    n = int(rng.integers(5, 400))
    beam = int(rng.integers(1, 120))
    topk = int(rng.integers(1, 40))
    if float(K) ** D > 4e18:
        continue
    w = synth.make_dr_model(n, K, D, L, E, rng, scale=float(rng.choice([0.05, 0.3, 1.0])))
    pi = synth.dr_path_items(synth.make_dr_paths(n, min(K, 6), D, 2, rng))
    eng = Engine(0)
    eng.dr_load_model(w, E, L, K, D, n, dtype=np.float64)
    eng.dr_load_path_items(*pi)
    orc = po.DeepRetrieval(w, E, L, K, D, n, path_items=pi)
    U = int(rng.integers(1, 8))
    seqs = rng.integers(0, n, size=(U, L)).astype(np.int32)
    seqs[rng.random((U, L)) < 0.3] = -1
    p, pr, cnt = eng.dr_beam_search(seqs, beam)
    ids, sc, rc = eng.dr_recommend(seqs, beam, topk)
    for u in range(U):
        op, ov = orc.beam_search(seqs[u], beam)
        oi, osc = orc.recommend(seqs[u], topk, beam)
        ok = (cnt[u] == len(op) and p[u, :cnt[u]].tolist() == op.tolist() and np.allclose(pr[u, :cnt[u]], ov, rtol=1e-9, atol=0)
              and rc[u] == len(oi) and ids[u, :rc[u]].tolist() == oi.tolist() and np.allclose(sc[u, :rc[u]], osc, rtol=1e-9, atol=1e-12))
        if not ok:
            bad += 1
            print("MISMATCH", c, dict(E=E, K=K, D=D, L=L, n=n, beam=beam, topk=topk, u=u))
            break
    eng.close()
print("configs", n_cfg, "mismatches", bad)
sys.exit(1 if bad else 0)
