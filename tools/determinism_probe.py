import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import random_din_weights, random_histories, synthetic_tree
from dismember_amd import Engine
rng = np.random.default_rng(77)
t = synthetic_tree(rng, 8, 200)
w = random_din_weights(rng, 128, 511)
eng = Engine(0)
eng.load_tree(t["codes"], t["ids"], t["is_leaf"], 8); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"]); eng.load_weights_din(w, 128, 511)
seqs = random_histories(rng, t["leaf_ids"], 9, 10)
ref = None
for rep in range(6):
    ids, sc, cnt, tc, ts, tn = eng.tdm_beam_search_trace(seqs, 128, 200)
    if ref is None:
        ref = (ids.copy(), sc.copy(), tc.copy(), ts.copy(), tn.copy())
        continue
    for u in range(9):
        if not np.array_equal(tn[u], ref[4][u]): print(rep, u, "trace counts differ", tn[u], ref[4][u])
        for it in range(tn.shape[1]):
            n = tn[u, it]
            if not np.array_equal(tc[u, it, :n], ref[2][u, it, :n]): print(rep, u, it, "trace codes differ")
            d = np.flatnonzero(ts[u, it, :n] != ref[3][u, it, :n])
            if d.size: print(rep, u, it, "trace scores differ at", d[:10], ts[u, it, d[:5]], ref[3][u, it, d[:5]])
        if not np.array_equal(ids[u], ref[0][u]):
            d = np.flatnonzero(ids[u] != ref[0][u]); print(rep, u, "final ids differ at", d[:10], "scores equal:", np.array_equal(np.sort(sc[u]), np.sort(ref[1][u])))
ref = None
for rep in range(20):
    ids, sc, cnt = eng.tdm_beam_search(seqs, 128, 200)
    if ref is None:
        ref = (ids.copy(), sc.copy()); continue
    for u in range(9):
        if not np.array_equal(ids[u], ref[0][u]) or not np.array_equal(sc[u], ref[1][u]):
            d = np.flatnonzero(ids[u] != ref[0][u])
            same_set = np.array_equal(np.sort(sc[u]), np.sort(ref[1][u]))
            lut = dict(zip(ref[0][u].tolist(), ref[1][u].tolist()))
            bad = [(int(i), float(s), lut[int(i)]) for i, s in zip(ids[u], sc[u]) if lut[int(i)] != s]
            print("notrace", rep, u, "ids differ at", d[:8], "score multiset equal:", same_set, "per-id score mismatches:", bad[:4], "sorted desc:", bool((np.diff(sc[u]) <= 0).all()))
print("done")
