"""Timing of one training step (TDM, E=128, 1M-item tree, batch of 8192 expanded rows) on one GPU."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth
E, depth = 128, 20
NI = (1 << (depth + 1)) - 1
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(1_000_000, depth, rng)
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(E, NI, synth.SEED, tree_depth=depth, rho=0.9)
eng.train_init(lr=1e-4)
neg = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17, 19, 22, 25, 30], np.int32)   # configs/tdm.conf prefix
per = int(sum(1 + neg[l] for l in range(1, depth + 1)))
T = max(1, 8192 // per)
seqs = synth.make_users(tree["leaf_ids"], T, 10, rng); tgt = rng.choice(tree["leaf_ids"], T).astype(np.int32)
t0 = time.perf_counter(); codes, rs, mask, y = eng.make_train_batch(seqs, tgt, neg, 1, seed=1); t_sample = time.perf_counter() - t0
pad = eng.rowmask_to_flat(mask, 10)
print("targets", T, "rows", codes.size, "sample+expand ms", t_sample * 1e3)
for it in range(4):
    t0 = time.perf_counter(); loss = eng.train_forward_backward(codes, rs, pad, y); t1 = time.perf_counter()
    eng.adam_step(); t2 = time.perf_counter()
    print("step %d: loss %.4f  fwd/bwd %.2f ms (incl. H2D)  adam+refresh %.2f ms" % (it, loss, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
