#!/usr/bin/env python3
"""recall@200 of the beam search against brute force on a TRAINED scorer at scale (round-5 verdict, next #7).

1M-item depth-20 tree (BASELINE configs[1]); TDMTrainer steps (level-wise negatives on the device, DIN fwd+bwd, Adam) on
tree-consistent synthetic interactions (synth.make_tree_consistent_interactions); at each checkpoint: beam-search top-k vs
dm_tdm_bruteforce_topk under the same weights, and the rate at which the held-out target itself is retrieved."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(eng, seqs, tgt, beam, topk):
    ids, sc, cnt = eng.tdm_beam_search(seqs, beam, topk)
    bids, _, bcnt = eng.tdm_bruteforce_topk(seqs, topk)
    rec = np.mean([len(set(ids[u, :cnt[u]].tolist()) & set(bids[u, :bcnt[u]].tolist())) / float(topk) for u in range(len(seqs))])
    hit = np.mean([int(tgt[u]) in set(ids[u, :cnt[u]].tolist()) for u in range(len(seqs))])
    bhit = np.mean([int(tgt[u]) in set(bids[u, :bcnt[u]].tolist()) for u in range(len(seqs))])
    return float(rec), float(hit), float(bhit)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--embed", type=int, default=128)
    ap.add_argument("--seq-len", type=int, default=10)
    ap.add_argument("--beam", type=int, default=200)
    ap.add_argument("--topk", type=int, default=200)
    ap.add_argument("--targets", type=int, default=256, help="targets per step")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--spread", type=float, default=64.0)
    ap.add_argument("--rho", type=float, default=0.95)
    ap.add_argument("--checkpoints", default="0,100,300,1000,2000")
    ap.add_argument("--eval-users", type=int, default=512)
    ap.add_argument("--neg", default="", help="comma list of per-level negative counts (default: configs/c2 conf)")
    a = ap.parse_args()
    from dismember_amd import Engine, synth, conf as dmconf
    from dismember_amd.trainer import TDMTrainer
    params = dmconf.task_params("TDMTrainDeepModel", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "c2_tdm_serve_1m.conf"))
    neg = np.array([int(x) for x in a.neg.split(",")] if a.neg else params["layer_negative_counts_list"], np.int32)
    E, L = a.embed, a.seq_len
    ni = (1 << (a.depth + 1)) - 1
    tree = synth.make_tree(a.items, a.depth, np.random.default_rng(synth.SEED))
    eng = Engine(0)
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], a.depth)
    eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    eng.load_weights_din_synthetic(E, ni, synth.SEED, tree_depth=a.depth, rho=a.rho)
    tr = TDMTrainer(eng, neg, lr=a.lr, seed=synth.SEED, sampler="device", with_prob=params["sample_with_probability"])
    ev_seq, ev_tgt = synth.make_tree_consistent_interactions(tree["leaf_ids"], a.eval_users, L, np.random.default_rng(synth.SEED + 991), a.spread)
    trng = np.random.default_rng(synth.SEED + 17)
    cps = sorted(int(x) for x in a.checkpoints.split(","))
    done, out = 0, []
    t_train = 0.0
    for cp in cps:
        t0 = time.perf_counter()
        loss = None
        while done < cp:
            s_, t_ = synth.make_tree_consistent_interactions(tree["leaf_ids"], a.targets, L, trng, a.spread)
            loss = tr.step(s_, t_)
            done += 1
        eng.synchronize()
        t_train += time.perf_counter() - t0
        rec, hit, bhit = measure(eng, ev_seq, ev_tgt, a.beam, a.topk)
        out.append({"steps": done, "loss": loss, "recall_at_k_vs_bruteforce": rec, "target_in_beam_topk": hit, "target_in_bruteforce_topk": bhit, "train_s": t_train})
        print(json.dumps(out[-1]), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
