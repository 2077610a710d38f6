"""Debug probe: per-phase cycle breakdown (thread 0 of every workgroup) of dr_beam_kernel; needs a -DDM_PHASE_TIMERS build
of the library passed as argv[1]."""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
from dismember_amd import Engine, synth
U = 16384
E, L, K, D, beam = 128, 10, 1000, 3, 50
rng = np.random.default_rng(synth.SEED)
eng = Engine(0)
eng.dr_load_model_synthetic(E, L, K, D, 1_000_000, synth.SEED, rerank=False, dtype=np.float64 if (len(sys.argv) > 2 and sys.argv[2] == 'f64') else np.float32)
seqs = rng.integers(0, 1_000_000, size=(U, L)).astype(np.int32)
d_seq = eng.dev_alloc(U * L * 4); eng.h2d(d_seq, seqs)
d_paths = eng.dev_alloc(U * beam * D * 4); d_probs = eng.dev_alloc(U * beam * 8); d_cnt = eng.dev_alloc(U * 4)
out = (C.c_ulonglong * 16)()
N.lib().dm_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
eng.dr_beam_search_dev(d_seq, U, beam, d_paths, d_probs, d_cnt); eng.synchronize()
N.lib().dm_debug_phase_cycles(eng._h, out)
eng.timing_reset()
eng.dr_beam_search_dev(d_seq, U, beam, d_paths, d_probs, d_cnt); eng.synchronize()
N.lib().dm_debug_phase_cycles(eng._h, out)
v = np.array(list(out)[:8], dtype=np.float64)
names = ["Srow load", "phase A (softmax stats + block maxima)", "block-maximum select", "candidate collection", "candidate sort", "next-state write", "output"]
print("kernel ms", eng.timing_get())
for n, x in zip(names, v): print("%-40s %6.2f%%  %8.0f cycles/user" % (n, 100 * x / v.sum(), x / U))
