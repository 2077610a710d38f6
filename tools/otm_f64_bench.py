"""Throughput of the fp64 OTM beam search (dm_otm_beam_search_f64: the reference's arithmetic for OTM) beside the f32 / split kernels on the
same model: complete depth-D tree, E = 128, beam 200.  Device-resident request (dm_otm_beam_search_dev); kernel time by HIP events.
  python tools/otm_f64_bench.py [depth=16] [users=16384] [pipeline=0]"""
import os, sys, time, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
U = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
if len(sys.argv) > 3 and sys.argv[3] == "1":
    os.environ["DM_OTM64_PIPELINE"] = "1"
from helpers import random_din_weights
from dismember_amd import Engine
E, L, beam = 128, 10, 200
NI = (1 << (depth + 1)) - 1
rng = np.random.default_rng(4)
first = (1 << depth) - 1
seqs = (first + rng.integers(0, 1 << depth, size=(U, L))).astype(np.int32)
seqs[rng.random((U, L)) < 0.15] = -1
eng = Engine(0)
if depth <= 18:
    w = random_din_weights(rng, E, NI).astype(np.float64)
    eng.load_weights_din(w, E, NI)
else:
    eng.load_weights_din_synthetic_f64(E, NI, 20250523)
d_seq = eng.dev_alloc(seqs.nbytes); eng.h2d(d_seq, seqs)
d_ids = eng.dev_alloc(U * 2 * beam * 4); d_sc = eng.dev_alloc(U * 2 * beam * 4); d_cnt = eng.dev_alloc(U * 4)
rows_per_user = 2 * (1 << (beam.bit_length() - 1)) + (depth - beam.bit_length()) * 2 * beam
modes = ("f64",) if (len(sys.argv) > 3 and sys.argv[3] == "f64only") else ("f64", "f32", "split_f16")
for mode in modes:
    eng.set_scorer_mode(mode)
    eng.otm_beam_search_dev(d_seq, min(U, 1024), L, beam, depth, d_ids, d_sc, d_cnt); eng.synchronize()
    eng.timing_reset()
    t0 = time.perf_counter()
    eng.otm_beam_search_dev(d_seq, U, L, beam, depth, d_ids, d_sc, d_cnt); eng.synchronize()
    dt = time.perf_counter() - t0
    n, ms = eng.timing_get()
    rows = eng.last_scored_rows()
    fl = rows * 2.0 * (E * E + 2 * L * E + E)
    print(json.dumps({"mode": mode, "kernel": eng.last_beam_kernel(), "users": U, "depth": depth, "wall_ms": round(dt * 1e3, 2), "kernel_ms": round(ms, 3), "launches": n,
                      "users_per_s": round(U / dt), "scored_rows": rows, "rows_per_user": rows / U if rows else rows_per_user,
                      "algorithmic_tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms else None,
                      "frac_of_fp64_peak_78.6": round(fl / (ms * 1e-3) / 78.6e12, 3) if (ms and mode == "f64") else None}))
eng.close()
