"""Randomised sweep of the two split-fp16 history GEMMs of the Deep-Retrieval search (GPU, f32 models): the 256 x 256 kernel over pre-split
operands (dr_gemm_split_x_kernel, forced onto every batch size) must return the 128 x 128 kernel's search results bit for bit.
  python tools/fuzz_dr_gemm_x.py [configs] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dismember_amd import Engine, synth
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(n_cfg):
    E = int(rng.choice([64, 128]))
    K = int(rng.integers(2, 400))
    D = int(rng.integers(2, 4))
    L = int(rng.integers(1, 25))
    n = int(rng.integers(5, 3000))
    beam = int(rng.integers(1, 60))
    U = int(rng.choice([1, 7, 255, 256, 257, 600, 1500]))
    w = synth.make_dr_model(n, K, D, L, E, rng, scale=float(rng.choice([0.05, 0.3])))
    seqs = rng.integers(0, n, size=(U, L)).astype(np.int32)
    seqs[rng.random((U, L)) < 0.3] = -1
    out = []
    for x in ("0", "1"):
        os.environ["DM_DR_GEMM_X"] = x
        os.environ["DM_DR_GEMM_X_MIN_ROWS"] = "1"
        eng = Engine(0)
        eng.dr_load_model(w, E, L, K, D, n, dtype=np.float32)
        out.append(eng.dr_beam_search(seqs, beam))
        eng.close()
    if not all(np.array_equal(a, b) for a, b in zip(*out)):
        bad += 1
        print("MISMATCH", c, dict(E=E, K=K, D=D, L=L, n=n, beam=beam, U=U))
print("configs %d, mismatches %d" % (n_cfg, bad))
sys.exit(1 if bad else 0)
