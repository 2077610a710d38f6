"""JTM.optimize at catalogue scale (BASELINE configs[3] shape on one GPU): n items x 4 training rows, depth-d tree, gap 2;
then the same gap steps once more with per-step timers (scoring vs re-balance).
  python tools/jtm_bench.py [items=1000000] [depth=20]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth
from dismember_amd.jtm import JTM
from dismember_amd import _native as N
items, depth = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 20)
E, L, nrow = 128, 10, 4
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(items, depth, rng)
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, synth.SEED, tree_depth=depth, rho=0.95)
t0 = time.perf_counter()
hist = synth.make_users(tree["leaf_ids"], 4 * 65536, L, np.random.default_rng(1))
order = np.argsort(tree["leaf_ids"], kind="stable")
pick = np.random.default_rng(2).integers(0, len(hist), size=items * nrow)
jt = JTM.from_arrays(eng, tree["leaf_ids"][order], tree["leaf_codes"][order], depth, np.arange(items + 1, dtype=np.int64) * nrow,
                     hist[pick].reshape(-1), gap=2, seq_len=L)
print("host preparation %.1f s" % (time.perf_counter() - t0))
tim = {}
t0 = time.perf_counter()
proj = jt.optimize(timing=tim, as_array=True)
dt = time.perf_counter() - t0
print("JTM.optimize: %d items, depth %d, %d gap steps: %.2f s  (%.0f items/s); bijection onto leaves: %s; split %s" %
      (items, depth, (depth + 1) // 2, dt, items / dt, np.unique(proj).size == items and int(proj.min()) >= (1 << depth) - 1,
       {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tim.items()}))
tw = tr = 0.0
proj = np.zeros(jt.items.size, np.int32)
eng._chk(N.lib().dm_jtm_cache_rows(eng._h, jt.row_off.ctypes.data_as(N.i64p), jt.row_ids.ctypes.data_as(N.i32p), jt.items.size, L)); jt._cached = True
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
