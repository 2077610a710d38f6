"""Throughput of the fp64 OTM beam search (dm_otm_beam_search_f64: the reference's arithmetic for OTM) beside the f32 / split kernels on the
same model: complete depth-16 tree (131 071 nodes), E = 128, beam 200, host-buffer entry points."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import random_din_weights
from dismember_amd import Engine
depth, E, L, beam, U = 16, 128, 10, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NI = (1 << (depth + 1)) - 1
rng = np.random.default_rng(4)
w = random_din_weights(rng, E, NI).astype(np.float64)
first = (1 << depth) - 1
seqs = (first + rng.integers(0, 1 << depth, size=(U, L))).astype(np.int32)
seqs[rng.random((U, L)) < 0.15] = -1
eng = Engine(0)
eng.load_weights_din(w, E, NI)
for mode in ("f64", "f32", "split_f16"):
    eng.set_scorer_mode(mode)
    eng.otm_beam_search(seqs[:64], beam, depth)
    t0 = time.perf_counter()
    ids, sc, cnt = eng.otm_beam_search(seqs, beam, depth)
    dt = time.perf_counter() - t0
    print("%-10s %8.0f users/s  (%d users, %.1f ms)" % (mode, U / dt, U, dt * 1e3))
eng.close()
