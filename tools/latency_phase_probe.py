"""Where the single-request latency goes inside the beam kernel (bundled E=16 model, one user, beam 20, topk 10): needs a
DM_PHASE_TIMERS probe build (tools/build_probe.sh DM_PHASE_TIMERS) as argv[1]; prints microseconds per phase of the one team."""
import ctypes as C, sys, os, numpy as np
os.environ["DM_TIME_DIRECT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
from dismember_amd import Engine
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
t = np.load(os.path.join(g, "tdm_tree.npz")); w = np.load(os.path.join(g, "din_f32.npy"))
eng = Engine(0)
eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
eng.load_weights_din(w, 16, 8191)
q = np.array([[0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882]], np.int32)
out = (C.c_ulonglong * 16)()
N.lib().dm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for _ in range(5): eng.tdm_beam_search(q, 20, 10)
N.lib().dm_debug_phase_cycles(eng._h, out)
eng.timing_reset()
R = 50
for _ in range(R): eng.tdm_beam_search(q, 20, 10)
n, ms = eng.timing_get()
N.lib().dm_debug_phase_cycles(eng._h, out)
v = np.array(list(out)[:12], dtype=np.float64) / R
names = ["setup: frontier init", "P3 expand (+barrier)", "P4 scoring", "P4 tail wait", "final select + user fetch", "user fetch -> setup(1)", "setup seq + K gather", "setup T1 + G", "P1 / no-sort", "P2 keygen", "P2 reg_sort", "P2 store + barrier"]
print("kernel %.1f us per call; the counters sum the cycles of ALL waves of the workgroup that reach a phase (idle teams too)" % (ms / n * 1e3))
for nm, x in zip(names, v): print("%-28s %9.0f cycles" % (nm, x))
