"""Kernel time of the TDM beam search on the 1M-item tree for a given build of the library (argv[1]) and scorer mode
(DM_SCORER=f32|split_f16): the A/B harness for kernel experiments."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
from dismember_amd import Engine, synth
U = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(1_000_000, 20, rng)
seqs = synth.make_users(tree["leaf_ids"], U, 10, np.random.default_rng(1))
eng = Engine(0)
eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], 20); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
eng.load_weights_din_synthetic(128, (1 << 21) - 1, synth.SEED, tree_depth=20, rho=0.95)
eng.set_scorer_mode(os.environ.get("DM_SCORER", "f32"))
d_seq = eng.dev_alloc(U * 40); d_ids = eng.dev_alloc(U * 800); d_sc = eng.dev_alloc(U * 800); d_cnt = eng.dev_alloc(U * 4)
eng.h2d(d_seq, seqs)
eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt); eng.synchronize()
eng.timing_reset()
for _ in range(3):
    eng.tdm_beam_search_dev(d_seq, U, 10, 200, 200, d_ids, d_sc, d_cnt)
eng.synchronize()
n, ms = eng.timing_get()
ids = np.empty((U, 200), np.int32); eng.d2h(ids, d_ids)
print("%-40s %s  %.3f ms/launch  %.0f users/s  checksum %d" % (os.path.basename(sys.argv[1]), os.environ.get("DM_SCORER", "f32"), ms / n, U * n / (ms * 1e-3), int(ids.astype(np.int64).sum())))
