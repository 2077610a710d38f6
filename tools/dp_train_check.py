"""2 workers on ONE GPU over gloo: data-parallel trainer replicas must stay bit-identical and match a single worker
that sees both slices (syncGradients semantics: mean of the workers' mean-loss gradients)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
from helpers import random_histories
from dismember_amd import Engine, sharding
from dismember_amd.trainer import TDMTrainer, exchange_gradients, EngineGradPort
import torch
dist, rank, world, local = sharding.init_distributed("gloo")
t = np.load("tests/golden/tdm_tree.npz"); w = np.load("tests/golden/din_f32.npy") * 0.3
def mk():
    e = Engine(0); e.load_tree(t["codes"], t["ids"], t["is_leaf"], 12); e.load_id_maps(t["leaf_ids"], t["leaf_codes"]); e.load_weights_din(w, 16, 8191); return e
rng = np.random.default_rng(0)
seqs = random_histories(rng, t["leaf_ids"], 64, 10); tgt = rng.choice(t["leaf_ids"], 64).astype(np.int32)
neg = np.arange(13, dtype=np.int32)
eng = mk(); eng.train_init(lr=1e-3)
class CpuPort(EngineGradPort):      # gloo cannot reduce device tensors: stage through host
    def dense(self): return super().dense().cpu()
    def set_dense(self, x): super().set_dense(x.cuda())
    def export_rows(self): r, g = super().export_rows(); return r.cpu(), g.cpu()
    def add_rows(self, r, g): super().add_rows(r.cuda(), g.cuda())
port = CpuPort(eng, torch)
lo, hi = sharding.shard_range(64, rank, world)
for it in range(3):
    c, s, m, y = eng.make_train_batch(seqs[lo:hi], tgt[lo:hi], neg, 1, seed=10 + it * 7 + rank)
    eng.train_forward_backward(c, s, eng.rowmask_to_flat(m, 10), y)
    n = exchange_gradients(port, dist, torch)
    eng.adam_step(1.0 / n)
wv = eng.train_download("weights")
allw = [None] * world
dist.all_gather_object(allw, wv)
if rank == 0:
    same = all(np.array_equal(allw[0], x) for x in allw[1:])
    # single worker emulating both: gradient = mean of the two workers' gradients
    ref = mk(); ref.train_init(lr=1e-3)
    for it in range(3):
        for r in range(world):
            a, b = sharding.shard_range(64, r, world)
            c, s, m, y = ref.make_train_batch(seqs[a:b], tgt[a:b], neg, 1, seed=10 + it * 7 + r)
            ref.train_forward_backward(c, s, ref.rowmask_to_flat(m, 10), y)
        ref.adam_step(1.0 / world)
    rw = ref.train_download("weights")
    d = np.abs(rw - allw[0])
    print("replicas bit-identical:", same, "| vs single-process accumulation: max |dw| =", float(d.max()), "(lr 1e-3)")
    assert same and d.max() < 2e-4
dist.barrier(); dist.destroy_process_group()
