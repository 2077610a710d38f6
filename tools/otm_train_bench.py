"""OTM training iteration at BASELINE configs[2] scale on one GPU: complete depth-24 tree over 10M items (33.5M nodes x 128),
beam 200: pseudo targets + beam nodes + one forward/backward + dense Adam per level."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth
from dismember_amd.otm_train import OTMTrainer
depth, E, L, beam = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), 128, 10, 200
U = int(sys.argv[2]) if len(sys.argv) > 2 else 20          # 20 users x 400 candidates = 8000 rows per level
eng = Engine(0)
ni = (1 << (depth + 1)) - 1
if len(sys.argv) > 3 and sys.argv[3] == "f64":          # the reference's DIN[Double]
    eng.load_weights_din_synthetic_f64(E, ni, synth.SEED)
else:
    eng.load_weights_din_synthetic(E, ni, synth.SEED, tree_depth=depth, rho=0.95)
rng = np.random.default_rng(3)
first = (1 << depth) - 1
seqs = (first + rng.integers(0, 1 << depth, size=(U, L))).astype(np.int32)
seqs[rng.random((U, L)) < 0.15] = -1
targets = [(first + rng.integers(0, 1 << depth, size=2)).tolist() for _ in range(U)]
t0 = time.perf_counter(); tr = OTMTrainer(eng, depth, beam, seq_len=L, lr=1e-4); eng.synchronize()
print("train_init (grad + Adam state for %d parameters): %.2f s" % (ni * E, time.perf_counter() - t0))
tr.train_batch(seqs, targets)
t0 = time.perf_counter(); losses = tr.train_batch(seqs, targets); eng.synchronize(); dt = time.perf_counter() - t0
print("OTM train_batch: %d users, %d levels (one Adam step each): %.2f s; losses %.4f .. %.4f" % (U, len(losses), dt, losses[0], losses[-1]))
print(tr.last_stats())
