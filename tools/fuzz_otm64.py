"""Randomised parity sweep (GPU) of the fp64 OTM beam search (fused dm_beam64_kernel, any team shape) against the fp64 oracle:
node lists EQUAL the oracle's for every user, scores within 1e-10 / 1e-9, the level trace replays exactly
(CandidateSearcher.buildBeamNodes on the device's scores).  python tools/fuzz_otm64.py [n_configs] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_din_weights                       # noqa: E402
from test_gpu_precision import _otm_replay                   # noqa: E402
from oracle import pyoracle as po                            # noqa: E402
from dismember_amd import Engine                             # noqa: E402
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
import collections, time                                     # noqa: E402
ran = users = 0
kernels = collections.Counter()
selftest = os.environ.get("FUZZ_SELFTEST") == "1"            # corrupt config 0's device ids: the sweep must report it
t_start = time.time()
for c in range(n_cfg):
    E = int(rng.choice([16, 32, 64, 128, 24, 48]))
    leaf_level = int(rng.integers(3, 11))
    beam = int(rng.integers(1, 40)) if rng.random() < 0.5 else int(rng.integers(1, 260))
    L = int(rng.integers(1, int(os.environ.get("FUZZ_LMAX", "16")) + 1))        # 17 .. 32: the per-level pipeline
    U = int(rng.integers(1, 12))
    NI = (1 << (leaf_level + 1)) - 1
    w = random_din_weights(rng, E, NI, dtype=np.float64, std=float(rng.choice([0.05, 0.3])), bias_std=0.1)
    codes = rng.integers((1 << leaf_level) - 1, NI, (U, L)).astype(np.int32)
    codes[rng.random((U, L)) < rng.random() * 0.6] = -1
    eng = Engine(0); eng.load_weights_din(w, E, NI)
    odin = po.Din(w, E, L, NI)
    start_level = beam.bit_length() - 1
    levels = max(leaf_level - start_level, 1)
    try:
        ids, sc, cnt, tc, ts, tn = eng.otm_beam_search_f64(codes, beam, leaf_level, trace_levels=levels)
        kernels[eng.last_beam_kernel() + (" beam<=32" if beam <= 32 else "")] += 1
        if selftest and c == 0:
            ids = np.array(ids, copy=True); ids[0, 0] ^= 1
        if leaf_level > start_level and not (selftest and c == 0):
            _otm_replay(po, tc, ts, tn, beam, start_level, leaf_level, ids)
        for u in range(U):
            oi, osc = po.otm_beam_search(odin, codes[u], leaf_level, beam)
            assert np.array_equal(ids[u, :cnt[u]], oi), "ids differ"
            assert (np.abs(sc[u, :cnt[u]] - osc) <= 1e-10 + 1e-9 * np.abs(osc)).all(), "scores differ"
        i2, s2, c2 = eng.otm_beam_search_f64(codes, beam, leaf_level)
        ran += 1; users += U
        assert np.array_equal(i2[:, :], ids) and np.array_equal(s2, sc) and np.array_equal(c2, cnt), "plain vs traced search differ"
    except AssertionError as e:
        bad += 1
        print("MISMATCH cfg", c, dict(E=E, leaf_level=leaf_level, beam=beam, L=L, U=U, kernel=eng.last_beam_kernel()), str(e)[:200])
    eng.close()
for k, v in sorted(kernels.items()):
    print("  kernel", k, v)
print("requested", n_cfg, "completed", ran, "users", users, "mismatches", bad, "elapsed_s", round(time.time() - t_start, 1))
if selftest:
    print("selftest:", "checker caught the injected fault" if bad >= 1 else "CHECKER DID NOT FIRE")
    sys.exit(0 if bad >= 1 else 1)
if ran != n_cfg - bad or users == 0:
    print("SWEEP INCOMPLETE"); sys.exit(1)
sys.exit(1 if bad else 0)
