"""The reference's command-line tasks over its `.conf` files, on the device path.

    python -m dismember_amd.tasks TDMInitializeTree --tdmConfFile configs/c1_tdm_movielens.conf
    python -m dismember_amd.tasks TDMTrainDeepModel --tdmConfFile configs/c1_tdm_movielens.conf [--quiet]
    python -m dismember_amd.tasks JTMTreeLearning   --jtmConfFile <file>
    python -m dismember_amd.tasks OTMTrainDeepModel --otmConfFile <file>     (then OTMConstructTree on the same file)

Each function is the body of the reference's `CommandApp` of the same name (examples/src/main/scala/com/mass/retrieval/):
the conf file is read by dismember_amd.conf (same keys, same defaults, same "failed to find <key>" stop), the steps are the
reference's, and every hot step is a library call — tree index + DIN weights in HBM, level-wise negative sampling, forward /
backward / Adam, the batched evaluator, `JTM.optimize` in one call.  What is NOT carried over: Java serialisation (the model file
is `dm_save_model`'s flat checkpoint), HDFS paths, the DeepFM graph and the k-means `TDMClusterTree` (out of scope).

  tdm_initialize_tree   tdm/TDMInitializeTree.scala:14-48  -> TreeInit.generate (tdm/.../tree/TreeInit.scala:33-66)
  tdm_train_deep_model  tdm/TDMTrainDeepModel.scala:20-83  -> LocalDataSet + LocalOptimizer.optimize (tdm/.../optim/LocalOptimizer.scala:58-126),
                        TDM.saveModel (:32-41), package.recommend (tdm/package.scala:114-124)
  jtm_tree_learning     jtm/JTMTreeLearning.scala:17-48    -> JTM.optimize (jtm/.../optim/JTM.scala:26-73), TreeUtil.writeTree
  otm_train_deep_model  otm/OTMTrainDeepModel.scala:20-75  -> LocalDataSet (dismember_amd/otm_data.py), LocalOptimizer.optimize, OTM.saveModel
  otm_construct_tree    otm/OTMConstructTree.scala:17-45   -> TreeConstruction.run, Serialization.saveMapping
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


def _otm_sample(path):
    """LocalDataSet.readFile (otm/.../dataset/LocalDataSet.scala:152-161) — category kept as read (the `category` leaf order compares it)"""
    if path.endswith(".npz"):
        return _read_interactions(path)
    users, items, cats, times = [], [], [], []
    with open(path) as f:
        for line in f:
            arr = line.strip().split(",")
            if len(arr) != 5 or not tree_io._is_creatable(arr[0]):
                continue
            users.append(int(arr[0])); items.append(int(arr[1])); times.append(int(arr[3])); cats.append(arr[4])
    return dict(user=users, item=items, category=cats, timestamp=times)


def otm_train_deep_model(conf_path, quiet=True, engine=None, max_iterations=None, time_recommend=True):
    """OTMTrainDeepModel (examples/.../otm/OTMTrainDeepModel.scala:20-75): data set + mapping (LocalDataSet), DIN[Double] over the
    complete tree of upperLog2(#items) levels, LocalOptimizer.optimize (otm/.../optim/LocalOptimizer.scala:55-110: per epoch a shuffle,
    batches of train_batch_size users, ONE library call per batch = pseudo targets + beam nodes + a gradient step per level), the
    evaluator at the progress interval and at the end of every epoch, OTM.saveModel (model checkpoint + `item node` mapping lines).
    -> dict(epoch_losses {"epoch k": per-level loss lists}, eval [(epoch, iter, loss, OtmEvalResult)], recommendation, mapping, params)"""
    from . import otm_data as od
    from .engine import Engine
    from .facade import OTM
    from .otm_train import OTMTrainer
    p = C.task_params("OTMTrainDeepModel", conf_path)
    if p["deep_model"] != "din":
        raise ValueError("DeepModel name should either be DeepFM or DIN (DeepFM is out of scope of this build)")
    if p["train_batch_size"] < p["thread_number"]:
        raise ValueError("requirement failed: train_batch_size >= thread_number")
    if not quiet:
        print("\n".join("%s: %s" % kv for kv in sorted(p.items())))
    L, E, beam = p["seq_len"], p["embed_size"], p["beam_size"]
    sample = _otm_sample(p["data_path"])
    rng = np.random.default_rng(p["seed"])
    if p["initialize_mapping"]:
        mapping = od.initialize_mapping(sample, p["leaf_init_mode"], rng)
    else:
        mapping = od.load_mapping(p["mapping_path"])
    consumed, train, evals = od.generate_samples(sample, mapping, L, p["min_seq_len"], p["split_ratio"], p["label_num"])
    leaf_level = od.upper_log2(len(mapping))
    ni = (1 << (leaf_level + 1)) - 1                                   # dataset.numTreeNode
    eng = engine or Engine(0)
    eng.load_weights_din(_din_init(E, ni, p["seed"], dtype=np.float64), E, ni)
    tr = OTMTrainer(eng, leaf_level=leaf_level, beam=beam, seq_len=L, lr=p["learning_rate"])
    allowed = ev.all_nodes(list(mapping.values()))
    tseq = np.array([t[0] for t in train], np.int32)
    ttgt = [t[1] for t in train]
    eseq = np.array([e[0] for e in evals], np.int32).reshape(-1, L)
    elab = [e[1] for e in evals]
    euser = np.array([e[2] for e in evals], np.int64)
    bs = p["train_batch_size"]
    n_batch = int(np.ceil(len(train) / float(bs)))
    interval = p["show_progress_interval"]
    epoch_losses, evals_out = {}, []
    done = 0
    for epoch in range(1, p["epoch_num"] + 1):
        order = rng.permutation(len(train))
        per_batch = []
        t_epoch, count = 0.0, 0
        for it in range(1, n_batch + 1):
            idx = order[(it - 1) * bs:it * bs]
            t0 = time.perf_counter()
            losses = tr.train_batch(tseq[idx], [ttgt[i] for i in idx], p["target_mode"])
            dt = time.perf_counter() - t0
            t_epoch += dt; count += len(idx)
            per_batch.append(losses)
            done += 1
            if it == n_batch or (interval > 0 and it % interval == 0):
                if len(evals):
                    loss, res = ev.evaluate_otm(eng, eseq, elab, euser, consumed, allowed, leaf_level, p["topk_number"], p["eval_batch_size"], beam)
                else:
                    loss, res = float("nan"), ev.OtmEvalResult()
                evals_out.append((epoch, it, loss, res))
                if not quiet:
                    print("Epoch %d Train %d/%d Iteration %d/%d Wall clock %.4fs Train time %.4fs Train loss %.4f\n\teval loss: %.4f, %s"
                          % (epoch, count, len(train), it, n_batch, t_epoch, dt, losses[-1], loss, res))
            if max_iterations is not None and done >= max_iterations:
                break
        epoch_losses["epoch %d" % epoch] = [list(x) for x in zip(*per_batch)]         # epochLoss.toList.transpose: per level, over the batches
        if max_iterations is not None and done >= max_iterations:
            break
    _mkdir_for(p["model_path"], p["mapping_path"])
    eng.save_model(p["model_path"])                                    # OTM.saveModel (OTM.scala:31-40)
    od.save_mapping(p["mapping_path"], mapping)
    otm = OTM(eng, mapping, p["deep_model"])
    query = [0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882]
    query = query[-L:] if L <= 10 else [0] * (L - 10) + query
    out = dict(epoch_losses=epoch_losses, eval=evals_out, mapping=mapping, params=p, engine=eng, n_train=len(train), n_eval=len(evals))
    out["recommendation"] = otm.recommend(query, 3, 20)
    if not quiet:
        print("Recommendation result: %s" % (out["recommendation"],))
    if time_recommend:                                                 # otm/package.scala:100-105
        for _ in range(10):
            otm.recommend(query, 10, 20)
        t0 = time.perf_counter()
        for _ in range(100):
            otm.recommend(query, 10, 20)
        out["recommend_ms"] = (time.perf_counter() - t0) * 10
        if not quiet:
            print("Average recommend time: %.4fms" % out["recommend_ms"])
    return out


def otm_construct_tree(conf_path, quiet=True, engine=None):
    """OTMConstructTree (examples/.../otm/OTMConstructTree.scala:17-45): the trained model + the mapping file -> TreeConstruction.run
    (otm/.../tree/TreeConstruction.scala:44-101) over the items' training histories -> the new `item node` mapping written back."""
    from . import otm_data as od
    from .engine import Engine
    from .otm_tree import TreeConstruction
    p = C.task_params("OTMConstructTree", conf_path)
    if not quiet:
        print("\n".join("%s: %s" % kv for kv in sorted(p.items())))
    eng = engine or Engine(0)
    eng.load_model(p["model_path"])
    mapping = od.load_mapping(p["mapping_path"])
    sample = _otm_sample(p["data_path"])
    seqs = od.item_sequences(sample, mapping, p["label_num"], p["min_seq_len"], p["seq_len"], p["split_ratio"])
    tc = TreeConstruction(eng, mapping, seqs, gap=p["gap"], seq_len=p["seq_len"], use_mask=p["use_mask"])
    t0 = time.perf_counter()
    result = tc.run()
    dt = time.perf_counter() - t0
    if not quiet:
        print("OTM tree construction time: %.4fs" % dt)
    od.save_mapping(p["mapping_path"], result)
    return dict(mapping=result, old_mapping=mapping, seconds=dt, params=p, engine=eng)


TASK_FUNCS = {
    "TDMInitializeTree": tdm_initialize_tree, "JTMInitializeTree": tdm_initialize_tree,
    "TDMTrainDeepModel": tdm_train_deep_model, "JTMTrainDeepModel": tdm_train_deep_model,
    "JTMTreeLearning": jtm_tree_learning,
    "OTMTrainDeepModel": otm_train_deep_model, "OTMConstructTree": otm_construct_tree,
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
