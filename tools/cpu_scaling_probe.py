import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import synth
from oracle import pyoracle as po
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(1_000_000, 20, rng)
w = synth.make_din_weights(128, (1 << 21) - 1, rng)
seqs = synth.make_users(tree["leaf_ids"], 4096, 10, np.random.default_rng(1))
otree = po.TdmTree(tree["codes"], tree["ids"], tree["is_leaf"], tree["leaf_ids"], tree["leaf_codes"], 20)
din = po.Din(w, 128, 10, (1 << 21) - 1)
os.system("lscpu | grep -i 'model name\\|socket\\|thread\\|numa node(s)\\|^CPU(s)'")
for nt in (1, 8, 32, 64, 128, 256):
    n = max(nt * 4, 8)
    t0 = time.perf_counter(); otree.recommend_batch(din, seqs[:n], 200, 200, n_threads=nt); dt = time.perf_counter() - t0
    print(nt, "threads:", n / dt, "users/s", n / dt / nt, "per thread")
