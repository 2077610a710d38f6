"""Debug probe: per-phase wave-cycle breakdown of the one-wave-per-SIMD beam kernel (beam_kernel_w.hip.inc); needs a
-DDM_PHASE_TIMERS build of the library passed as argv[1].  usage: phase_probe_w.py <lib.so> [users] [depth] [items]"""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
from dismember_amd import Engine, synth
U = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 24
items = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(items, depth, rng)
seqs = synth.make_users(tree["leaf_ids"], U, 10, np.random.default_rng(1))
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(128, (1 << (depth + 1)) - 1, synth.SEED, tree_depth=depth, rho=0.95)
d_seq = eng.dev_alloc(U * 40); d_ids = eng.dev_alloc(U * 800); d_sc = eng.dev_alloc(U * 800); d_cnt = eng.dev_alloc(U * 4)
eng.h2d(d_seq, seqs)
eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt); eng.synchronize()
out = (C.c_ulonglong * 16)()
N.lib().dm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
N.lib().dm_debug_phase_cycles(eng._h, out)
eng.timing_reset()
eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt); eng.synchronize()
N.lib().dm_debug_phase_cycles(eng._h, out)
v = np.array(list(out)[:8], dtype=np.float64)
names = ["per-user setup (K, T1, G, frontier)", "P1 + P2 prune (keys, register sort)", "P3 expand", "P4 prologue (first gather + split)",
         "P4 tiles", "P4 flush + level end", "user fetch / final selection", "-"]
n, ms = eng.timing_get()
ms /= max(n, 1)
rows = eng.last_scored_rows()
print("kernel ms %.2f, scored rows %d, users %d" % (ms, rows, U))
for nm, x in zip(names, v): print("%-40s %6.2f%%" % (nm, 100 * x / v.sum()))
n_waves = 256 * 4
clk = v.sum() / n_waves / (ms * 1e-3) / 1e9
print("shader clock under this kernel: %.3f GHz" % clk)
tiles = rows / 16.0
print("cycles per tile in the tile loop: %.0f (matrix pipe minimum 124 x 16 = 1984)" % (v[4] / tiles))
print("cycles per user outside the tile loop: %.0f" % ((v.sum() - v[4]) / U))
tv = np.array(list(out)[8:16], dtype=np.float64)
if tv.sum() > 0 and os.environ.get("DM_PROBE_SETUP"):
    print("setup sections (cycles per user, -DDM_SETUP_TIMERS): decode %.0f | K rows + split %.0f | T1 %.0f | G %.0f | G split %.0f | frontier %.0f"
          % tuple(tv[:6] / U))
elif tv.sum() > 0:
    print("tile sections (cycles per tile, -DDM_TILE_TIMERS): scores %.0f | W groups %s | PG %.0f | tail %.0f" %
          (tv[0] / tiles, " ".join("%.0f" % (x / tiles) for x in tv[1:3]), tv[3] / tiles, tv[4] / tiles))
