"""Single-request latency of TDM.recommend on the bundled model while other engines of the same process hold a large model, training
state, or have just run searches (round 6: the bench's configs[0] timer read 0.77 ms instead of 0.06 after the trained-recall extra)."""
import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from dismember_amd import Engine, TDM, synth
from dismember_amd.trainer import TDMTrainer
g = "/root/repo/tests/golden"
t = np.load(os.path.join(g, "tdm_tree.npz")); w = np.load(os.path.join(g, "din_f32.npy"))
def lat(tag):
    e1 = Engine(0)
    e1.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); e1.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    e1.load_weights_din(w, 16, 8191)
    m1 = TDM(e1, "din")
    q = np.array([0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882], np.int32)
    for _ in range(10): m1.recommend(q, 10, 20)
    t0 = time.perf_counter()
    for _ in range(100): m1.recommend(q, 10, 20)
    print(tag, "%.1f us" % ((time.perf_counter() - t0) / 100 * 1e6), flush=True)
    e1.close()
lat("fresh")
depth, items, E, L = 20, 1_000_000, 128, 10
tree = synth.make_tree(items, depth, np.random.default_rng(1))
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, 7, tree_depth=depth, rho=0.95)
lat("with a 1M-item engine alive")
from dismember_amd import conf as dmconf
neg = np.array(dmconf.task_params("TDMTrainDeepModel", "/root/repo/configs/c2_tdm_serve_1m.conf")["layer_negative_counts_list"], np.int32)
tr = TDMTrainer(eng, neg, lr=1e-3, seed=1, sampler="device")
lat("after train_init")
rng = np.random.default_rng(2)
for _ in range(20):
    s_, t_ = synth.make_tree_consistent_interactions(tree["leaf_ids"], 256, L, rng, 64.0)
    tr.step(s_, t_)
eng.synchronize()
lat("after 20 train steps")
ids = eng.tdm_beam_search(synth.make_users(tree["leaf_ids"], 512, L, rng), 200, 200)
lat("after a search on the trained engine")
b = eng.tdm_bruteforce_topk(synth.make_users(tree["leaf_ids"], 64, L, rng), 200)
lat("after brute force")
