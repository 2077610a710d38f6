"""How the headline kernel behaves on beams that DIVERGE (round-4 verdict, weak #5 / next #2).  With the reference's N(0, 0.05) init the
user-dependent term of the logit is several times smaller than the node term, so the beams of different users overlap heavily and the
gathers are served from L2 (96.8 % hit rate in the headline profile).  This tool re-runs the same search (same tree, same users) for

    * the headline model (rho 0.95 tree-correlated table, reference init),
    * the same table with the attention path scaled up (att.W and W1b x `s`) and, optionally, a larger embedding scale, so that the
      history decides the beam,
    * an iid table (rho 0: SURVEY.md §8d's literal table),

and prints, per variant: users/s (device-resident request, HIP events around the kernel), how much the beams of different users
overlap — distinct candidate rows per level over a 4 096-user trace, relative to (users x candidates) — and the distinct items in the
results.  `python tools/diverse_bench.py [users=131072] [steps=6] [variants=head,s8,s16e4,rho0]`; PMC passes (TCC hit rate, FETCH_SIZE) come
from running it under rocprofv3 (tools/collect_profiles_r05.sh diverse).
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
variants = (sys.argv[3] if len(sys.argv) > 3 else "head,s8,s16e4,rho0").split(",")
items, depth, E, L, beam, topk = 10_000_000, 24, 128, 10, 200, 200
NI = (1 << (depth + 1)) - 1
rng = np.random.default_rng(synth.SEED)
tree = synth.make_tree(items, depth, rng)
seqs = synth.make_users(tree["leaf_ids"], U, L, np.random.default_rng(synth.SEED + 1))


def small_matrices(s):
    r = np.random.default_rng(int(synth.SEED))
    small = np.zeros(3 * E * E + 2 * E + 1, np.float32)
    small[:3 * E * E] = r.standard_normal(3 * E * E, dtype=np.float32) * 0.05
    small[3 * E * E + E:3 * E * E + 2 * E] = r.standard_normal(E, dtype=np.float32) * 0.05
    small[:E * E] *= s                                              # att.W
    l1 = small[E * E:3 * E * E].reshape(E, 2 * E)
    l1[:, E:] *= s                                                  # W1b (the attention half of linear1)
    return small


def run(name):
    s, rho, escale = 1.0, 0.95, 1.0
    if name.startswith("s"):
        body = name[1:]
        if "e" in body:
            a_, b_ = body.split("e"); s, escale = float(a_), float(b_)
        else:
            s = float(body)
    elif name == "rho0":
        rho = 0.0
    eng = Engine(0)
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    # (a larger embedding scale sharpens the attention softmax, q.k / sqrt(E): the table is drawn with std 0.05 x escale)
    eng.load_weights_din_synthetic(E, NI, synth.SEED, small=small_matrices(s), tree_depth=depth, rho=rho, std=0.05 * escale)
    d_seq = eng.dev_alloc(U * L * 4); eng.h2d(d_seq, seqs)
    d_ids, d_sc, d_cnt = eng.dev_alloc(U * topk * 4), eng.dev_alloc(U * topk * 4), eng.dev_alloc(U * 4)
    eng.tdm_beam_search_dev(d_seq, U, L, beam, topk, d_ids, d_sc, d_cnt); eng.synchronize()
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.tdm_beam_search_dev(d_seq, U, L, beam, topk, d_ids, d_sc, d_cnt)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nl, kms = eng.timing_get_kind(0)
    nd, dms = eng.timing_get_kind(1)          # second pass over the users the one-wave kernel deferred (|G + b1| outside the fp16 range)
    rows = eng.last_scored_rows()
    ids = np.empty((U, topk), np.int32); eng.d2h(ids, d_ids)
    # beam overlap: trace of a 4 096-user sample, distinct candidate codes per level / (users x candidates of the level)
    Us = min(4096, U)
    tr = eng.tdm_beam_search_trace(seqs[:Us], beam, topk)
    tc, tn = tr[3], tr[5]
    per_level = []
    for lv in range(tn.shape[1]):
        n = tn[:, lv]
        if n.max() == 0:
            continue
        cand = np.concatenate([tc[u, lv, :n[u]] for u in range(Us)])
        per_level.append(round(float(np.unique(cand).size) / float(cand.size), 4))
    out = {"variant": name, "att_scale": s, "emb_scale": escale, "rho": rho, "users_per_s": U / dt, "ms_per_step": dt * 1e3,
           "kernel_ms": kms / max(nl, 1), "deferred_pass_ms": dms / max(nd, 1), "kernel": eng.last_beam_kernel(), "scored_rows_per_user": rows / U,
           "distinct_result_items": int(np.unique(ids[ids >= 0]).size), "result_slots": int((ids >= 0).sum()),
           "distinct_candidate_rows_per_level_fraction_4096_users": per_level,
           "mean_distinct_fraction_levels_12_up": float(np.mean(per_level[max(0, len(per_level) - 12):]))}
    print(json.dumps(out), flush=True)
    for p in (d_seq, d_ids, d_sc, d_cnt):
        eng.dev_free(p)
    eng.close()
    return out


if __name__ == "__main__":
    for v in variants:
        run(v)
