"""Single-request latency of the facade calls (the reference prints "Average recommend time": one user, topk 10, beam 20,
10 warm-up + 100 timed calls — examples/.../tdm/package.scala:119-123)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, TDM
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
t = np.load(os.path.join(g, "tdm_tree.npz")); w = np.load(os.path.join(g, "din_f32.npy"))
eng = Engine(0)
eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
eng.load_weights_din(w, 16, 8191)
tdm = TDM(eng, "din")
q = np.array([0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882], np.int32)
for _ in range(10):
    tdm.recommend(q, 10, 20)
t0 = time.perf_counter()
for _ in range(100):
    tdm.recommend(q, 10, 20)
print("TDM.recommend (1 user, topk 10, beam 20, E=16, depth 12): %.1f us per call" % ((time.perf_counter() - t0) / 100 * 1e6))
qb = np.tile(q, (256, 1))
for _ in range(3):
    tdm.recommend(qb, 10, 20)
t0 = time.perf_counter()
for _ in range(20):
    tdm.recommend(qb, 10, 20)
print("256 users per call: %.1f us per call" % ((time.perf_counter() - t0) / 20 * 1e6))
eng.close()
os.environ["DM_TIME_DIRECT"] = "1"          # the single-request path launches without its event pair unless asked
eng = Engine(0)
eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
eng.load_weights_din(w, 16, 8191)
tdm = TDM(eng, "din")
for _ in range(10):
    tdm.recommend(q, 10, 20)
eng.timing_reset()
for _ in range(100):
    tdm.recommend(q, 10, 20)
n, ms = eng.timing_get()
print("kernel time inside the single-user call: %.1f us (HIP events, %d launches)" % (ms / n * 1e3, n))
# ---- OTM.recommend on the bundled DIN[Double] (examples/.../otm/package.scala:101-105)
from dismember_amd import OTM
eng.close()
w6 = np.load(os.path.join(g, "din_f64.npy")); om = np.load(os.path.join(g, "otm_mapping.npy"))
e2 = Engine(0); e2.load_weights_din(w6, 16, 8191)
mo = OTM(e2, {int(a_): int(b_) for a_, b_ in om})
qo = [int(x) for x in om[:10, 0]]
for _ in range(10):
    mo.recommend(qo, 10, 20)
t0 = time.perf_counter()
for _ in range(100):
    mo.recommend(qo, 10, 20)
print("OTM.recommend (1 user, topk 10, beam 20, fp64): %.1f us per call (%s)" % ((time.perf_counter() - t0) / 100 * 1e6, e2.last_beam_kernel()))
e2.timing_reset()
for _ in range(100):
    mo.recommend(qo, 10, 20)
n, ms = e2.timing_get()
print("kernel time inside the OTM call: %.1f us (%d launches)" % (ms / max(n, 1) * 1e3, n))
