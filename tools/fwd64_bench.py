"""dm_din_forward on an f64 model, E = 128, L = 10: rows per second of the matrix-pipe forward (default for batches >= 256 rows) —
run again with DM_FWD64_SCALAR=1 for the one-wave-per-row kernel.  python tools/fwd64_bench.py [rows=1000000]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
E, L, depth = 128, 10, 20
NI = (1 << (depth + 1)) - 1
eng = Engine(0)
eng.load_weights_din_synthetic_f64(E, NI, 7)
rng = np.random.default_rng(1)
codes = rng.integers(0, NI, B).astype(np.int32)
seqs = rng.integers(0, NI, (B, L)).astype(np.int32); seqs[rng.random((B, L)) < 0.2] = -1
eng.din_forward(codes[:4096], seqs[:4096])
eng.synchronize()
t0 = time.perf_counter(); out = eng.din_forward(codes, seqs); eng.synchronize(); dt = time.perf_counter() - t0
print("%s: %d rows in %.1f ms (host buffers in and out) = %.1f M rows/s; finite %s" % ("scalar kernel" if os.environ.get("DM_FWD64_SCALAR") == "1" else "matrix pipe", B, dt * 1e3, B / dt / 1e6, bool(np.isfinite(out).all())))
