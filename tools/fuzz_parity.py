"""Randomised parity sweep (GPU): many small random configurations of the TDM search against the CPU oracle, through the
trace replay contract (ids bit-exact on the GPU's scores, scores within tolerance).  Not part of the test suite; run
by hand: python tools/fuzz_parity.py [n_configs] [seed]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_din_weights, random_histories, synthetic_tree    # noqa: E402
from test_gpu_parity import make_engine, replay_and_check                    # noqa: E402
from oracle import pyoracle as po                                            # noqa: E402

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
# what the sweep actually did (the summary line reports these, not the request): configs completed, users compared against the
# oracle, plain-entry calls small enough for the host-mapped path, kernel labels seen.  FUZZ_SELFTEST=1 corrupts the device
# result of config 0 before the check: the sweep must then report a mismatch (it proves the checker can fail).
import collections, time                                                     # noqa: E402
ran = users = small_u = 0
kernels = collections.Counter()
selftest = os.environ.get("FUZZ_SELFTEST") == "1"
t_start = time.time()
for c in range(n_cfg):
    E = int(rng.choice([16, 32, 64, 128]))
    depth = int(rng.integers(4, 12))
    n_items = int(rng.integers(max(2, (1 << depth) // 3), (1 << depth) + 1))
    beam = int(rng.integers(1, 300)) if rng.random() < 0.5 else int(rng.integers(1, 40))      # half the sweep on small beams: one-wave teams
    topk = int(rng.integers(1, 2 * beam + 2))
    U = int(rng.integers(1, 20))
    LMAX = int(os.environ.get("FUZZ_LMAX", "0"))                # 0: the configs' L = 10; n: L uniform in 1 .. n (17 .. 32 = the per-level pipeline)
    L = int(rng.integers(1, LMAX + 1)) if LMAX else 10
    t = synthetic_tree(rng, depth, n_items)
    NI = (1 << (depth + 1)) - 1
    w = random_din_weights(rng, E, NI)
    otree = po.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])
    odin = po.Din(w, E, L, NI)
    eng = make_engine(t, w, E)
    eng.set_scorer_mode(os.environ.get("DM_SCORER", "auto"))
    seqs = random_histories(rng, t["leaf_ids"], U, L, pad_prob=float(rng.random()) * 0.6, unknown_prob=0.05)
    try:
        um = bool(rng.integers(0, 2))
        replay_and_check(otree, odin, eng, seqs, beam, topk, use_mask=um)
        # the plain entry point (for <= 8 users: the host-mapped single-request path) must return what the traced search returned
        a = eng.tdm_beam_search_trace(seqs, beam, topk, use_mask=um)[:3]
        b = eng.tdm_beam_search(seqs, beam, topk, use_mask=um)
        kernels[eng.last_beam_kernel() + (" beam<=32" if beam <= 32 else "")] += 1
        if selftest and c == 0:
            b = tuple(np.array(x, copy=True) for x in b); b[0].flat[0] ^= 1
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), "plain vs traced search differ"
        if U <= 8:
            small_u += 1
        ran += 1; users += U
    except AssertionError as e:
        bad += 1
        print("MISMATCH cfg", c, dict(E=E, depth=depth, n_items=n_items, beam=beam, topk=topk, U=U, L=L), str(e)[:300])
    eng.close()
for k, v in sorted(kernels.items()):
    print("  kernel", k, v)
print("requested", n_cfg, "completed", ran, "users", users, "plain calls with U<=8", small_u, "mismatches", bad,
      "elapsed_s", round(time.time() - t_start, 1))
if selftest:
    print("selftest:", "checker caught the injected fault" if bad >= 1 else "CHECKER DID NOT FIRE")
    sys.exit(0 if bad >= 1 else 1)
if ran + bad != n_cfg or users == 0:
    print("SWEEP INCOMPLETE"); sys.exit(1)
sys.exit(1 if bad else 0)
