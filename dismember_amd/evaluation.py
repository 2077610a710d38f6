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
