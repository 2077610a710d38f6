"""The reference's command-line tasks over its `.conf` files, on the device path.

    python -m dismember_amd.tasks TDMInitializeTree --tdmConfFile configs/c1_tdm_movielens.conf
    python -m dismember_amd.tasks TDMTrainDeepModel --tdmConfFile configs/c1_tdm_movielens.conf [--quiet]
    python -m dismember_amd.tasks JTMTreeLearning   --jtmConfFile <file>

Each function is the body of the reference's `CommandApp` of the same name (examples/src/main/scala/com/mass/retrieval/):
the conf file is read by dismember_amd.conf (same keys, same defaults, same "failed to find <key>" stop), the steps are the
reference's, and every hot step is a library call — tree index + DIN weights in HBM, level-wise negative sampling, forward /
backward / Adam, the batched evaluator, `JTM.optimize` in one call.  What is NOT carried over: Java serialisation (the model file
is `dm_save_model`'s flat checkpoint), HDFS paths, the DeepFM graph and the k-means `TDMClusterTree` (out of scope).

  tdm_initialize_tree   tdm/TDMInitializeTree.scala:14-48  -> TreeInit.generate (tdm/.../tree/TreeInit.scala:33-66)
  tdm_train_deep_model  tdm/TDMTrainDeepModel.scala:20-83  -> LocalDataSet + LocalOptimizer.optimize (tdm/.../optim/LocalOptimizer.scala:58-126),
                        TDM.saveModel (:32-41), package.recommend (tdm/package.scala:114-124)
  jtm_tree_learning     jtm/JTMTreeLearning.scala:17-48    -> JTM.optimize (jtm/.../optim/JTM.scala:26-73), TreeUtil.writeTree
"""
import os
import sys
import time

import numpy as np

from . import conf as C
from . import evaluation as ev
from . import tree_io


def _mkdir_for(*paths):
    for p in paths:
        if p:
            d = os.path.dirname(p)
            if d:
                os.makedirs(d, exist_ok=True)


def _read_interactions(path):
    """TreeInit.readFile: `user,item,label,timestamp,category` lines.  The repository's fixture of the reference's bundled sample
    (tests/golden/example_data.npz: the decoded columns of data/example_data.csv in file order) is accepted as well."""
    if path.endswith(".npz"):
        d = np.load(path)
        item = d["item"].astype(int)
        cat_of = dict(zip(d["uniq_item"].astype(int).tolist(), d["uniq_cat"].astype(int).tolist()))
        return dict(user=d["user"].astype(int).tolist(), item=item.tolist(), category=[cat_of[i] for i in item.tolist()],
                    timestamp=list(range(item.size)))
    with open(path) as f:
        return tree_io.read_interactions(f)


def tdm_initialize_tree(conf_path, quiet=True):
    p = C.task_params("TDMInitializeTree", conf_path)
    if not quiet:
        print("\n".join("%s: %s" % kv for kv in sorted(p.items())))
    sample = _read_interactions(p["data_path"])
    split = tree_io.split_samples(tree_io.user_sequences(sample), p["seq_len"], p["min_seq_len"], p["split_for_eval"], p["split_ratio"])
    ids, codes, uniq = tree_io.gen_codes(sample["item"], sample["category"])
    _mkdir_for(p["train_path"], p["eval_path"], p["stat_path"], p["leaf_id_path"], p["tree_protobuf_path"], p["user_consumed_path"])
    tree_io.write_split_files(split, p["train_path"], p["eval_path"] if p["split_for_eval"] else None, p["stat_path"], p["user_consumed_path"])
    with open(p["leaf_id_path"], "w") as f:                  # TreeInit.initializeTree: the distinct item ids, one per line
        for i in uniq:
            f.write("%d\n" % i)
    tree_io.write_tree_file(p["tree_protobuf_path"], ids, codes, split["stat"])
    if not quiet:
        print("item num: %d, train samples: %d, eval samples: %d" % (len(ids), len(split["train"]), len(split["eval"])))
    return dict(n_items=len(ids), n_train=len(split["train"]), n_eval=len(split["eval"]), params=p)


def _din_init(E, num_index, seed, dtype=np.float32):
    """DIN.buildModel's initial parameters in A0 order: N(0, 0.05) tables and weights, zero biases
    (tdm/.../model/DIN.scala:12-42, scalann/.../nn/EmbeddingShare.scala:19-22)."""
    rng = np.random.default_rng(seed)
    n = num_index * E + E * E + 2 * E * E + E + E + 1
    w = (rng.standard_normal(n) * 0.05).astype(dtype)
    b1 = num_index * E + E * E + 2 * E * E
    w[b1:b1 + E] = 0
    w[-1] = 0
    return w


def tdm_train_deep_model(conf_path, quiet=True, engine=None, seed=2024, max_iterations=None, time_recommend=True):
    """-> dict(losses, eval (list of (iteration, EvalResult means)), recommendation, params)."""
    from .engine import Engine
    from .facade import TDM
    from .trainer import TDMTrainer
    p = C.task_params("TDMTrainDeepModel", conf_path)
    if p["deep_model"] != "din":
        raise ValueError("DeepModel name should either be DeepFM or DIN (DeepFM is out of scope of this build)")
    if not quiet:
        print("\n".join("%s: %s" % kv for kv in sorted(p.items())))
    L, E = p["seq_len"], p["embed_size"]
    eng = engine or Engine(0)
    eng.load_tree_file(p["tree_protobuf_path"])                       # TDMOp.initTree
    depth = eng.max_level
    ni = (1 << (depth + 1)) - 1
    eng.load_weights_din(_din_init(E, ni, seed), E, ni)
    with open(p["train_path"]) as f:
        tseq, ttgt = tree_io.read_train_data(f)
    with open(p["eval_path"]) as f:
        eseq, elab, euser = tree_io.read_eval_data(f, L)
    with open(p["user_consumed_path"]) as f:
        consumed = tree_io.read_user_consumed(f)
    neg = np.asarray(p["layer_negative_counts_list"][:depth + 1], np.int32)        # NegativeSampler.scala:55-57
    start = p["start_sample_level"]
    per = int(sum(1 + int(neg[l]) for l in range(start, depth + 1)))                # rows one target expands to (MiniBatch.scala:23-38)
    T = max(1, p["total_batch_size"] // per)                                       # expandBatch = true
    tr = TDMTrainer(eng, neg, start_level=start, use_mask=p["use_mask"], lr=p["learning_rate"], seed=seed,
                    with_prob=p["sample_with_probability"], tolerance=p["sample_tolerance"])
    rng = np.random.default_rng(seed)
    order = rng.permutation(len(ttgt))                                             # dataset.shuffle()
    n_iter = p["iteration_number"] if max_iterations is None else min(max_iterations, p["iteration_number"])
    interval = p["show_progress_interval"]
    losses, evals = [], []
    pos, epoch, count, t_epoch = 0, 0, 0, 0.0

    def report(it, loss, dt):
        m = ev.evaluate(eng, eseq, elab, euser, consumed, neg, topk=p["topk_number"], candidate_num=p["beam_size"],
                        use_mask=p["use_mask"], batch_size=p["total_eval_batch_size"], start_level=start, seed=seed + 7)
        evals.append((it, m.means()))
        if not quiet:
            print("Epoch %d Train %d/%d Iteration %d Wall clock %.4fs Train time %.4fs Train loss %.4f\n\t%s"
                  % (epoch + 1, count, len(ttgt), it, t_epoch, dt, loss, m))

    for it in range(1, n_iter + 1):
        t0 = time.perf_counter()
        idx = order[pos:pos + T]
        loss = tr.step(tseq[idx], ttgt[idx])
        dt = time.perf_counter() - t0
        t_epoch += dt
        losses.append(float(loss))
        pos += len(idx); count += len(idx)
        if interval > 0 and it % interval == 0:
            report(it, loss, dt)
        if count >= len(ttgt):                                                     # LocalOptimizer.scala:98-116
            report(it, loss, dt)
            epoch += 1
            order = rng.permutation(len(ttgt)); pos, count, t_epoch = 0, 0, 0.0
    _mkdir_for(p["model_path"], p["embed_path"])
    tdm = TDM(eng, p["deep_model"])
    tdm.save_model(p["model_path"])
    out = dict(losses=losses, eval=evals, params=p, engine=eng)
    query = [0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882][-L:] if L <= 10 else [0] * (L - 10) + [0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882]
    rec = tdm.recommend(query, 3, 20)
    out["recommendation"] = rec
    if not quiet:
        print("Recommendation result: %s" % (rec,))
    if time_recommend:                                                             # tdm/package.scala:118-123
        for _ in range(10):
            tdm.recommend(query, 10, 20)
        t0 = time.perf_counter()
        for _ in range(100):
            tdm.recommend(query, 10, 20)
        out["recommend_ms"] = (time.perf_counter() - t0) * 10
        if not quiet:
            print("Average recommend time: %.4fms" % out["recommend_ms"])
    return out


def jtm_tree_learning(conf_path, quiet=True, engine=None):
    """JTMTreeLearning: the model of `model_path` and the tree of `tree_protobuf_path` -> JTM.optimize over the training rows of
    `data_path` -> the re-assigned tree written back to `tree_protobuf_path` (TreeUtil.writeTree, jtm/.../tree/JTMTree.scala:115-182)."""
    from .engine import Engine
    from .jtm import JTM
    p = C.task_params("JTMTreeLearning", conf_path)
    if not quiet:
        print("\n".join("%s: %s" % kv for kv in sorted(p.items())))
    eng = engine or Engine(0)
    eng.load_model(p["model_path"])
    eng.load_tree_file(p["tree_protobuf_path"])
    with open(p["tree_protobuf_path"], "rb") as f:
        t = tree_io.read_tree_bytes(f.read())
    rows = {}
    with open(p["data_path"]) as f:                                                # TreeLearning.readDataFile (TreeLearning.scala:34-46)
        for line in f:
            arr = line.strip().split(",")
            if len(arr) < 3:
                continue
            rows.setdefault(int(arr[-1]), []).extend(int(float(x)) for x in arr[1:-1])
    rows = {k: np.asarray(v, np.int32) for k, v in rows.items()}
    jtm = JTM(eng, t["leaf_ids"], t["leaf_codes"], t["max_level"], rows, gap=p["gap"], seq_len=p["seq_len"],
              hierarchical=p["hierarchical_preference"], min_level=p["min_level"], use_mask=p["use_mask"])
    t0 = time.perf_counter()
    proj = jtm.optimize()
    dt = time.perf_counter() - t0
    if not quiet:
        print("JTM tree learning time: %.4fs" % dt)
    items = np.array(sorted(proj), np.int32)
    codes = np.array([proj[int(i)] for i in items], np.int32)
    code_prob = dict(zip(t["codes"].tolist(), t["probs"].tolist()))                # the probability an item's OLD leaf node carried
    old_code = dict(zip(np.asarray(t["leaf_ids"]).tolist(), np.asarray(t["leaf_codes"]).tolist()))
    probs = np.array([code_prob.get(old_code[int(i)], 0.0) for i in items], np.float32)
    non_leaf_offset = int(np.asarray(t["leaf_ids"]).max()) + 1                      # DistTree.scala:35
    with open(p["tree_protobuf_path"], "wb") as f:
        f.write(tree_io.build_jtm_tree_bytes(items, codes, probs, t["max_level"], non_leaf_offset))
    return dict(projection=proj, seconds=dt, params=p, engine=eng)


TASK_FUNCS = {
    "TDMInitializeTree": tdm_initialize_tree, "JTMInitializeTree": tdm_initialize_tree,
    "TDMTrainDeepModel": tdm_train_deep_model, "JTMTrainDeepModel": tdm_train_deep_model,
    "JTMTreeLearning": jtm_tree_learning,
}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in C.TASKS:
        print("usage: python -m dismember_amd.tasks <%s> --<x>ConfFile <file> [--quiet]" % " | ".join(sorted(C.TASKS)), file=sys.stderr)
        return 2
    task = argv[0]
    flag = C.TASKS[task][0]
    quiet = "--quiet" in argv
    path = None
    for i, a in enumerate(argv):
        if a == "--" + flag and i + 1 < len(argv):
            path = argv[i + 1]
        elif a.startswith("--" + flag + "="):
            path = a.split("=", 1)[1]
    if path is None:
        print("missing --%s <file> (the reference's `fromResource` default has no counterpart here)" % flag, file=sys.stderr)
        return 2
    if task not in TASK_FUNCS:
        print("%s: out of scope of this build (SURVEY.md §2); its conf keys are read by dismember_amd.conf.task_params" % task, file=sys.stderr)
        return 3
    TASK_FUNCS[task](path, quiet=quiet)
    return 0


if __name__ == "__main__":
    sys.exit(main())
