"""Deep-Retrieval in the reference's arithmetic type (fp64): D=3, K=1000, beam=50, E=128 on a 200k-item catalogue
(the work per user does not depend on the catalogue size)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth
E, L, K, D, n, beam, U = 128, 10, 1000, 3, 200_000, 50, 8192
rng = np.random.default_rng(1)
w = synth.make_dr_model(n, K, D, L, E, rng, scale=0.05)
for k in ("rerank_emb", "rerank_w", "rerank_b", "softmax_w", "softmax_b"):
    w[k] = None
seqs = rng.integers(0, n, size=(U, L)).astype(np.int32)
for dt in (np.float64, np.float32):
    eng = Engine(0)
    eng.dr_load_model(w, E, L, K, D, n, dtype=dt)
    d_seq = eng.dev_alloc(U * L * 4); eng.h2d(d_seq, seqs)
    d_p = eng.dev_alloc(U * beam * D * 4); d_pr = eng.dev_alloc(U * beam * 8); d_c = eng.dev_alloc(U * 4)
    eng.dr_beam_search_dev(d_seq, U, beam, d_p, d_pr, d_c); eng.synchronize(); eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.dr_beam_search_dev(d_seq, U, beam, d_p, d_pr, d_c)
    eng.synchronize()
    dtm = (time.perf_counter() - t0) / 3
    print(np.dtype(dt).name, "users/s %.0f" % (U / dtm), "ms/step %.2f" % (dtm * 1e3), "kernel ms", eng.timing_get())
    eng.close()
