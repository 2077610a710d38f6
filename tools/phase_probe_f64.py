"""Where a wave's time goes inside dm_beam64_kernel (fp64 OTM beam): needs a DM64_PHASE_TIMERS probe build
(bash tools/build_probe.sh DM64_PHASE_TIMERS) as argv[1].  python tools/phase_probe_f64.py <lib> [depth=16] [users=32768]"""
import ctypes as C, os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dismember_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
from helpers import random_din_weights
from dismember_amd import Engine
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
U = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
E, L, beam = 128, 10, 200
NI = (1 << (depth + 1)) - 1
rng = np.random.default_rng(4)
first = (1 << depth) - 1
seqs = (first + rng.integers(0, 1 << depth, size=(U, L))).astype(np.int32)
seqs[rng.random((U, L)) < 0.15] = -1
eng = Engine(0)
eng.load_weights_din(random_din_weights(rng, E, NI).astype(np.float64), E, NI)
d_seq = eng.dev_alloc(seqs.nbytes); eng.h2d(d_seq, seqs)
d_ids = eng.dev_alloc(U * 2 * beam * 4); d_sc = eng.dev_alloc(U * 2 * beam * 4); d_cnt = eng.dev_alloc(U * 4)
out = (C.c_ulonglong * 16)()
N.lib().dm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
eng.set_scorer_mode("f64")
eng.otm_beam_search_dev(d_seq, 1024, L, beam, depth, d_ids, d_sc, d_cnt); eng.synchronize()
N.lib().dm_debug_phase_cycles(eng._h, out)
eng.timing_reset()
eng.otm_beam_search_dev(d_seq, U, L, beam, depth, d_ids, d_sc, d_cnt); eng.synchronize()
n, ms = eng.timing_get()
N.lib().dm_debug_phase_cycles(eng._h, out)
v = np.array(list(out)[:8], dtype=np.float64)
sub = np.array(list(out)[8:12], dtype=np.float64)
names = ["setup", "prune", "tiles", "team-barrier waits", "level tail", "output", "user fetch", "-"]
print(json.dumps({"kernel": eng.last_beam_kernel(), "kernel_ms": ms, "users": U, "depth": depth,
                  "shares": {k: round(float(x / v.sum()), 4) for k, x in zip(names, v) if x}, "counter_total": float(v.sum()),
                  "tile_split": {k: round(float(x / max(v[2], 1)), 4) for k, x in zip(["scores (K fragment loads + 32 MFMAs)", "softmax", "W1a chain + PG + relu.w2"], sub)}}))
eng.close()
