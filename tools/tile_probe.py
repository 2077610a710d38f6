"""Debug probe: where a wave's time goes INSIDE the scoring tile loop (needs a -DDM_TILE_TIMERS build as argv[1])."""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
from dismember_amd import Engine, synth
U = 32768
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(1_000_000, 20, rng)
seqs = synth.make_users(tree["leaf_ids"], U, 10, np.random.default_rng(1))
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], 20); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(128, (1 << 21) - 1, synth.SEED, tree_depth=20, rho=0.95)
eng.set_scorer_mode(os.environ.get("DM_SCORER", "f32"))
d_seq = eng.dev_alloc(U * 40); d_ids = eng.dev_alloc(U * 800); d_sc = eng.dev_alloc(U * 800); d_cnt = eng.dev_alloc(U * 4)
eng.h2d(d_seq, seqs)
out = (C.c_ulonglong * 16)()
N.lib().dm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt); eng.synchronize()
N.lib().dm_debug_phase_cycles(eng._h, out)
eng.timing_reset()
eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt); eng.synchronize()
N.lib().dm_debug_phase_cycles(eng._h, out)
v = np.array(list(out)[:5], dtype=np.float64)
rows = eng.last_scored_rows()
tiles = rows / 16.0
names = ["S^T MFMAs + next-tile gather issue", "softmax", "main chain (256 MFMAs)", "P x G (24 MFMAs)", "epilogue + store"]
print("kernel ms", eng.timing_get(), "tiles", tiles)
for n, x in zip(names, v): print("%-40s %6.2f%%  %8.0f clock64 ticks / tile" % (n, 100 * x / v.sum(), x / tiles))
print("total ticks / tile %.0f" % (v.sum() / tiles))
