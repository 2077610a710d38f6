"""JTM.optimize at catalogue scale (BASELINE configs[3] shape on one GPU): n items x 4 training rows, depth-d tree, gap 2."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth
from dismember_amd.jtm import JTM
items, depth = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 20)
E, L = 128, 10
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(items, depth, rng)
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, synth.SEED, tree_depth=depth, rho=0.95)
t0 = time.perf_counter()
hist = synth.make_users(tree["leaf_ids"], 4 * 65536, L, np.random.default_rng(1))
pick = np.random.default_rng(2).integers(0, len(hist), size=(items, 4))
rows = {int(it): hist[pick[k]].reshape(-1) for k, it in enumerate(tree["leaf_ids"])}
jt = JTM(eng, tree["leaf_ids"], tree["leaf_codes"], depth, rows, gap=2, seq_len=L)
print("host preparation %.1f s" % (time.perf_counter() - t0))
t0 = time.perf_counter()
proj = jt.optimize()
dt = time.perf_counter() - t0
codes = np.fromiter(proj.values(), np.int64)
print("JTM.optimize: %d items, depth %d, %d gap steps: %.2f s  (%.0f items/s); bijection onto leaves: %s" %
      (items, depth, (depth + 1) // 2, dt, items / dt, len(set(codes.tolist())) == items and codes.min() >= (1 << depth) - 1))
# where the time goes: scoring vs re-balance per gap step
import ctypes as C
from dismember_amd import _native as N
tw = tr = 0.0
proj = np.zeros(jt.items.size, np.int32)
for old_level in range(0, depth, 2):
    level = min(depth, old_level + 2)
    t0 = time.perf_counter(); w = jt.child_weights(proj, old_level, level); t1 = time.perf_counter()
    old_node = JTM.ancestor_at_level(jt.item_code, level)
    new = np.empty_like(proj)
    t2 = time.perf_counter()
    eng._chk(N.lib().dm_jtm_rebalance_all(eng._h, w.ctypes.data_as(N.f32p), old_node.ctypes.data_as(N.i32p), proj.ctypes.data_as(N.i32p),
                                          proj.size, old_level, level, 1 << (depth - level), new.ctypes.data_as(N.i32p)))
    t3 = time.perf_counter()
    tw += t1 - t0; tr += t3 - t2
    print("  levels %2d -> %2d: scoring %.2f s, re-balance %.2f s (%d parent nodes)" % (old_level, level, t1 - t0, t3 - t2, np.unique(proj).size), flush=True)
    proj = new
print("scoring %.2f s, re-balance %.2f s" % (tw, tr))
