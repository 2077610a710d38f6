"""Brute-force mode rate: the beam kernel's scoring phase almost alone (pool maintenance is tiny) — an upper
bound for what the level loop's P4 can reach on the MFMA pipe."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import _native as _N
if os.environ.get("DM_LIB"): _N.LIB_PATH = os.path.abspath(os.environ["DM_LIB"])
from dismember_amd import Engine, synth
E, L, depth, items = 128, 10, 20, 1_000_000
U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(items, depth, rng)
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, synth.SEED, tree_depth=depth, rho=0.95)
seqs = synth.make_users(tree["leaf_ids"], U, L, np.random.default_rng(1))
eng.tdm_bruteforce_topk(seqs[:8], 200)
eng.timing_reset()
t0 = time.perf_counter()
eng.tdm_bruteforce_topk(seqs, 200)
dt = time.perf_counter() - t0
nl, ms = eng.timing_get()
rows = U * items
print("users", U, "wall s", dt, "kernel ms", ms, "launches", nl, "TFLOP/s (38144/row)", rows * 38144 / (ms * 1e-3) / 1e12,
      "frac", rows * 38144 / (ms * 1e-3) / 1e12 / 157.3)
