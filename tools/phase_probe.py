"""Debug probe: per-phase wave-cycle breakdown of the beam kernel (needs a -DDM_PHASE_TIMERS build
passed as argv[1])."""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
from dismember_amd import Engine, synth
U = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(1_000_000, 20, rng)
w = synth.make_din_weights(128, (1 << 21) - 1, rng)
seqs = synth.make_users(tree["leaf_ids"], U, 10, np.random.default_rng(1))
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], 20); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"]); eng.load_weights_din(w, 128, (1 << 21) - 1)
eng.set_scorer_mode(os.environ.get("DM_SCORER", "f32"))
d_seq = eng.dev_alloc(U * 40); d_ids = eng.dev_alloc(U * 800); d_sc = eng.dev_alloc(U * 800); d_cnt = eng.dev_alloc(U * 4)
eng.h2d(d_seq, seqs)
eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt); eng.synchronize()
out = (C.c_ulonglong * 16)()
N.lib().dm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
N.lib().dm_debug_phase_cycles(eng._h, out)
eng.timing_reset()
eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt); eng.synchronize()
N.lib().dm_debug_phase_cycles(eng._h, out)
v = np.array(list(out)[:12], dtype=np.float64)
names = ["setup: frontier init (+G/K frag write)", "P3 expand (+barrier)", "P4 scoring (own tiles)", "P4 tail wait (barrier)", "final select + user fetch", "user fetch barrier->setup(1)", "setup (1)-(2) seq+K gather", "setup (3)-(4) T1+G", "P1 + (no-sort path)", "P2 keygen", "P2 reg_sort", "P2 store + barrier"]
print("kernel ms", eng.timing_get())
for n, x in zip(names, v): print("%-28s %6.2f%%" % (n, 100 * x / v.sum()))
n_waves = 256 * 8
ms = eng.timing_get()[1] / max(eng.timing_get()[0], 1)
print("shader clock under this kernel: %.3f GHz (sum of per-wave clock64 ticks / waves / kernel time)" % (v.sum() / n_waves / (ms * 1e-3) / 1e9))
