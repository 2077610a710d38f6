"""Evaluator + metrics: the reference's `Evaluator.evaluate`
(tdm/src/main/scala/com/mass/tdm/evaluation/Evaluator.scala:14-74), `Metrics.computeMetrics` (Metrics.scala:5-25) and
`EvalResult` (EvalResult.scala:3-37) over the device entry points: one batched `dm_tdm_beam_search` (consumed-item
filtering + the beam-widening rule of Recommender.scala:28-33) per eval batch instead of one recommend call per
user, and one `dm_din_forward` for the loss rows.  Only the O(topk) set intersections and the BCE reduction of the
logits run on the host."""
import numpy as np

_LOG2 = np.log(2.0)


def compute_metrics(rec_items, labels):
    """(precision, recall, ndcg) of one user: precision over the items actually returned, ideal DCG over as many
    leading positions as there are hits (Metrics.scala:5-25)."""
    rec = np.asarray(rec_items)
    k = rec.size
    if k == 0:
        return (0.0, 0.0, 0.0)
    hit = np.isin(rec, np.asarray(labels))
    common = int(hit.sum())
    if common == 0:
        return (0.0, 0.0, 0.0)
    pos = np.flatnonzero(hit)
    # sequential accumulation in position order, as the reference's while loop
    dcg = idcg = 0.0
    for j, i in enumerate(pos):
        dcg += _LOG2 / np.log(i + 2.0)
        idcg += _LOG2 / np.log(j + 2.0)
    return (common / float(k), common / float(len(labels)), dcg / idcg)


class EvalResult:
    """Running sums (EvalResult.scala:3-37); str() prints the reference's line."""

    def __init__(self, loss=0.0, precision=0.0, recall=0.0, ndcg=0.0, count=0):
        self.loss, self.precision, self.recall, self.ndcg, self.count = loss, precision, recall, ndcg, count

    def __add__(self, o):
        self.loss += o.loss; self.precision += o.precision; self.recall += o.recall; self.ndcg += o.ndcg; self.count += o.count
        return self

    def add_metrics(self, v):
        self.precision += v[0]; self.recall += v[1]; self.ndcg += v[2]

    def means(self):
        c = float(self.count)
        return dict(loss=self.loss / c, precision=self.precision / c, recall=self.recall / c, ndcg=self.ndcg / c)

    def __str__(self):
        m = self.means()
        return "{eval loss: %.4f, precision: %.6f, recall: %.6f, ndcg: %.6f}" % (m["loss"], m["precision"], m["recall"], m["ndcg"])


def bce_with_logits(logits, targets):
    """BCECriterionWithLogits.updateOutput, sizeAverage (scalann/.../nn/BCECriterionWithLogits.scala:27-64), float32."""
    x = np.asarray(logits, np.float32)
    z = np.asarray(targets, np.float32)
    buf = np.maximum(x, np.float32(0)) + np.log(np.float32(1) + np.exp(-np.abs(x)))
    return float((buf.sum(dtype=np.float32) - np.dot(x, z)) / np.float32(x.size))


def evaluate(engine, sequences, labels, users, user_consumed, neg_counts, topk, candidate_num, use_mask=True,
             batch_size=8192, start_level=1, seed=0, return_batches=False):
    """sequences [N, L] item ids, labels: list of N id arrays (target = labels[i][0], TDMSample.scala:31-38),
    users [N], user_consumed: dict user -> ids.  One worker per call (the reference splits an eval batch over
    Engine.coreNumber() model clones and sums `mean loss x length` per clone)."""
    seqs = np.ascontiguousarray(sequences, np.int32)
    N, L = seqs.shape
    neg = np.asarray(neg_counts, np.int32)
    per = int(sum(1 + int(neg[l]) for l in range(start_level, engine.max_level + 1)))      # MiniBatch.scala:23-38
    step = max(1, batch_size // per)
    total = EvalResult()
    batches = []
    for off in range(0, N, step):
        n = min(step, N - off)
        tgt = np.array([labels[i][0] for i in range(off, off + n)], np.int32)
        codes, rseq, rmask, rlab = engine.make_train_batch(seqs[off:off + n], tgt, neg, start_level=start_level,
                                                           seed=seed + off, use_mask=use_mask)
        pad = engine.rowmask_to_flat(rmask, L)
        out = engine.din_forward(codes, rseq, pad, L=L)
        res = EvalResult(loss=bce_with_logits(out, rlab) * n, count=n)
        ids, _, cnt = engine.tdm_beam_search(seqs[off:off + n], candidate_num, topk, use_mask=use_mask,
                                             consumed=[user_consumed[int(users[i])] for i in range(off, off + n)],
                                             widen_consumed=True)
        for i in range(n):
            res.add_metrics(compute_metrics(ids[i, :cnt[i]], labels[off + i]))
        total = total + res
        if return_batches:
            batches.append((off, n, codes, rseq, pad, rlab))
    return (total, batches) if return_batches else total


# ----------------------------------------------------------------------------------------------------------------------
# OTM evaluator: otm/src/main/scala/com/mass/otm/evaluation/Evaluator.scala:29-90 (+ Metrics.scala:7-31, EvalResult.scala:3-27)

def all_nodes(leaf_node_ids):
    """LocalDataSet.getAllNodes (otm/.../dataset/LocalDataSet.scala:199-205): the mapped leaves and their ancestors, as a set."""
    ids = np.asarray(list(leaf_node_ids), np.int64)
    leaf_level = int(np.ceil(np.log(ids.size) / np.log(2)))            # upperLog2, otm/package.scala:16
    out = set(ids.tolist())
    a = ids.copy()
    for _ in range(leaf_level):
        a = (a - 1) // 2                 # (Scala's Int division truncates toward zero: (0 - 1) / 2 = 0)
        a[a < 0] = 0
        out.update(a.tolist())
    return out


def bce_with_logits_sum(logits, targets):
    """BCECriterionWithLogits(sizeAverage = false).updateOutput in double (BCECriterionWithLogits.scala:28-60):
    sum(max(x, 0) + log(1 + exp(-|x|))) - x . z"""
    x = np.asarray(logits, np.float64)
    z = np.asarray(targets, np.float64)
    if x.size == 0:
        return 0.0
    buf = np.maximum(x, 0.0) + np.log(np.exp(-np.abs(x)) + 1.0)
    return float(buf.sum() - np.dot(x, z))


class OtmEvalResult:
    """otm/.../evaluation/EvalResult.scala:3-27"""

    def __init__(self, precision=0.0, recall=0.0, ndcg=0.0):
        self.precision, self.recall, self.ndcg = precision, recall, ndcg

    def __add__(self, o):
        return OtmEvalResult(self.precision + o.precision, self.recall + o.recall, self.ndcg + o.ndcg)

    def __truediv__(self, size):
        return OtmEvalResult(self.precision / size, self.recall / size, self.ndcg / size)

    def __str__(self):
        return "{precision: %.6f, recall: %.6f, ndcg: %.6f}" % (self.precision, self.recall, self.ndcg)


def evaluate_otm(engine, sequences, labels, users, user_consumed, all_node_ids, leaf_level, topk, total_eval_batch_size,
                 beam_size, thread_num=1, f64=None):
    """Evaluator.evaluate (Evaluator.scala:29-84) over the device search: one batched OTM beam search per eval batch
    (CandidateSearcher.batchBeamSearch), then per user — drop consumed node ids, keep mapped nodes (dataset.allNodes), stable
    sort by score descending, take(topk) (:56-61); labels of the kept nodes (:63-68), Metrics.computeMetrics (:69), one
    BCE-with-logits SUM per worker chunk (:75-77, computeLoss :86-96).  Everything is in node-id space, as in the reference's
    OTM data set.  Returns (total loss / eval size, OtmEvalResult / eval size).

    sequences [N, L] node ids (-1 = padding), labels: list of N node-id lists (OTMSample.labels), users [N],
    user_consumed: dict user -> node ids; thread_num: the reference's Engine.coreNumber() (it only groups the loss sums);
    f64: search in the reference's double precision (default: when the engine holds f64 weights)."""
    seqs = np.ascontiguousarray(sequences, np.int32)
    N = seqs.shape[0]
    if f64 is None:
        f64 = engine.scorer_mode()["mode"] == "f64"
    allowed = all_node_ids if isinstance(all_node_ids, (set, frozenset)) else set(np.asarray(all_node_ids).tolist())
    batch = max(1, total_eval_batch_size // (beam_size * 2))
    total_loss, total = 0.0, OtmEvalResult()
    for off in range(0, N, batch):
        n = min(batch, N - off)
        if f64:
            ids, sc, cnt = engine.otm_beam_search_f64(seqs[off:off + n], beam_size, leaf_level)[:3]
        else:
            ids, sc, cnt = engine.otm_beam_search(seqs[off:off + n], beam_size, leaf_level)
        chunk = int(np.ceil(n / float(thread_num)))
        for c0 in range(0, n, chunk):
            preds, labs = [], []
            for i in range(c0, min(n, c0 + chunk)):
                consumed = set(np.asarray(user_consumed[int(users[off + i])]).tolist())
                nodes, scores = ids[i, :cnt[i]], np.asarray(sc[i, :cnt[i]], np.float64)
                keep = np.array([(int(v) not in consumed) and (int(v) in allowed) for v in nodes], bool)
                nodes, scores = nodes[keep], scores[keep]
                order = np.argsort(-scores, kind="stable")[:topk]
                nodes, scores = nodes[order], scores[order]
                tgt = labels[off + i]
                tset = set(int(t) for t in tgt)
                preds.append(scores)
                labs.append(np.array([1.0 if int(v) in tset else 0.0 for v in nodes], np.float64))
                m = compute_metrics(nodes, list(tgt))
                total = total + OtmEvalResult(*m)
            total_loss += bce_with_logits_sum(np.concatenate(preds) if preds else [], np.concatenate(labs) if labs else [])
    return total_loss / N, total / N


# ----------------------------------------------------------------------------------------------------------------------
# Deep-Retrieval evaluator: deep-retrieval/src/main/scala/com/mass/dr/evaluation/Evaluator.scala:15-130

class DrEvalResult:
    """deep-retrieval/.../evaluation/EvalResult.scala:7-57: layer / re-rank losses averaged over the batches, metrics over samples."""

    def __init__(self, layer_loss, rerank_loss, precision, recall, ndcg, size, count=1):
        self.layer_loss, self.rerank_loss = list(layer_loss), rerank_loss
        self.precision, self.recall, self.ndcg, self.size, self.count = precision, recall, ndcg, size, count

    def __add__(self, o):
        return DrEvalResult([a + b for a, b in zip(self.layer_loss, o.layer_loss)], self.rerank_loss + o.rerank_loss,
                            self.precision + o.precision, self.recall + o.recall, self.ndcg + o.ndcg, self.size + o.size,
                            self.count + o.count)

    def mean_metrics(self):
        return DrEvalResult([v / self.count for v in self.layer_loss], self.rerank_loss / self.count, self.precision / self.size,
                            self.recall / self.size, self.ndcg / self.size, self.size, self.count)

    def __str__(self):
        def fmt(v):                      # DecimalFormat("##.####")
            s = ("%.4f" % v).rstrip("0").rstrip(".")
            return s if s not in ("", "-") else "0"
        return ("eval layer loss: [%s], rerank loss: %.4f\n\t\tprecision: %.6f, recall: %.6f, ndcg: %.6f"
                % (", ".join(fmt(v / self.count) for v in self.layer_loss), self.rerank_loss / self.count,
                   self.precision / self.size, self.recall / self.size, self.ndcg / self.size))


def evaluate_dr(engine, sequences, labels, users, user_consumed, topk, beam_size, batch_size=8192, num_layer=0, loss_fn=None):
    """Evaluator.evaluate (Evaluator.scala:15-73): per eval mini-batch the recommendation metrics of every sample —
    recommendItems (:108-129): candidate items of the beam's paths, consumed items dropped, re-rank scores, stable sort
    descending, take(topk) — summed with Metrics.computeMetrics.  The device call re-ranks BEFORE the consumed filter, so it is
    asked for topk + |consumed| items; dropping consumed ids from that list leaves exactly the reference's list (both sorts are
    stable over the same candidate order).  Ids are the model's internal item ids (MappingOp.itemIdMapping), as in the reference's
    data set.

    The two loss columns (evaluateLayerModel :75-85, evaluateReRankModel :87-97) belong to Deep-Retrieval TRAINING, which is out
    of scope here (config 5 is serving-only): `loss_fn(off, n) -> (layer_losses, rerank_loss)` supplies them when
    a caller has the training-side modules, otherwise they are reported as 0 over `num_layer` layers."""
    seqs = np.ascontiguousarray(sequences, np.int32)
    N = seqs.shape[0]
    total = None
    for off in range(0, N, batch_size):
        n = min(batch_size, N - off)
        cons = [set(np.asarray(user_consumed[int(users[off + i])]).tolist()) for i in range(n)]
        k = min(2048, topk + max((len(c) for c in cons), default=0))
        ids, _, cnt = engine.dr_recommend(seqs[off:off + n], beam_size, k)
        p = r = g = 0.0
        for i in range(n):
            rec = [int(v) for v in ids[i, :cnt[i]] if int(v) not in cons[i]][:topk]
            if len(rec) < topk and cnt[i] == k and k < topk + len(cons[i]):
                # the device list was cut at dm_dr_recommend's 2 048 items before `topk` unconsumed ones were seen: say so instead of
                # silently scoring a shorter list than the reference would
                raise ValueError("evaluate_dr: user %d has %d consumed items; topk + consumed exceeds dm_dr_recommend's 2048-item limit"
                                 % (int(users[off + i]), len(cons[i])))
            m = compute_metrics(np.asarray(rec, np.int64), list(labels[off + i]))
            p += m[0]; r += m[1]; g += m[2]
        ll, rl = loss_fn(off, n) if loss_fn else ([0.0] * num_layer, 0.0)
        res = DrEvalResult(ll, rl, p, r, g, n)
        total = res if total is None else total + res
    return total
