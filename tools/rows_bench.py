"""Throughput of the general-rows DIN forward through the host ABI (includes H2D/D2H; indicative only)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth
E, depth = 128, 20
NI = (1 << (depth + 1)) - 1
eng = Engine(0)
eng.load_weights_din_synthetic(E, NI, synth.SEED, tree_depth=depth, rho=0.9)
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
codes = rng.integers(0, NI, B).astype(np.int32)
seqs = rng.integers(0, NI, (B, 10)).astype(np.int32)
seqs[rng.random((B, 10)) < 0.1] = -1
pad = np.flatnonzero(seqs.reshape(-1) == -1).astype(np.int32)
for _ in range(3):
    t0 = time.perf_counter(); out = eng.din_forward(codes, seqs, pad); dt = time.perf_counter() - t0
    print("rows/s incl. transfers: %.3g   (%.1f ms)" % (B / dt, dt * 1e3))
