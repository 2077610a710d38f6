"""TEST INFRASTRUCTURE — restatement of the reference's evaluator, pure Python loops
(small inputs only).  T/ = /root/reference/tdm/src/main/scala/com/mass/tdm/.

  compute_metrics  T/evaluation/Metrics.scala:5-25
  EvalResult       T/evaluation/EvalResult.scala:3-37
  bce_with_logits  scalann/.../nn/BCECriterionWithLogits.scala:27-64 (sizeAverage = true)
  evaluate         T/evaluation/Evaluator.scala:14-74 with one worker (Engine.coreNumber() = 1); the negative-sampled loss
                   batches are passed in (the reference's sampler is unseeded)
Parity status: pinned on hand-computed known answers (tests/test_evaluation.py); the reference's own test
(tdm/src/test/scala/TdmModelTrainSpec.scala) only prints these metrics.
"""
import math

import numpy as np


def compute_metrics(rec_items, labels):
    k = len(rec_items)
    label_set = set(int(x) for x in labels)
    i = j = common = 0
    dcg = idcg = 0.0
    while i < k:
        if int(rec_items[i]) in label_set:
            common += 1
            dcg += math.log(2) / math.log(i + 2)
            idcg += math.log(2) / math.log(j + 2)
            j += 1
        i += 1
    if common != 0:
        return (common / float(k), common / float(len(labels)), dcg / idcg)
    return (0.0, 0.0, 0.0)


class EvalResult:
    def __init__(self, loss=0.0, precision=0.0, recall=0.0, ndcg=0.0, count=0):
        self.loss, self.precision, self.recall, self.ndcg, self.count = loss, precision, recall, ndcg, count

    def __add__(self, o):
        self.loss += o.loss; self.precision += o.precision; self.recall += o.recall; self.ndcg += o.ndcg; self.count += o.count
        return self

    def add_metrics(self, v):
        self.precision += v[0]; self.recall += v[1]; self.ndcg += v[2]

    def __str__(self):
        c = self.count
        return "{eval loss: %.4f, precision: %.6f, recall: %.6f, ndcg: %.6f}" % (self.loss / c, self.precision / c,
                                                                              self.recall / c, self.ndcg / c)


def bce_with_logits(x, z):
    """max(x,0) + log(1 + exp(-|x|)) summed, minus dot(x, z), divided by the batch size; float32 like the reference."""
    x = np.asarray(x, np.float32)
    z = np.asarray(z, np.float32)
    a = np.float32(0)
    b = np.float32(0)
    for xi, zi in zip(x, z):
        a = np.float32(a + (np.maximum(xi, np.float32(0)) + np.log(np.float32(1) + np.exp(-np.abs(xi)))))
        b = np.float32(b + xi * zi)
    return float(np.float32(a - b) / np.float32(len(x)))


def evaluate(tree, din, sequences, labels, users, user_consumed, loss_batches, topk, candidate_num, use_mask=True):
    """loss_batches: list of (offset, length, codes, seqs, pad_flat, row_labels) covering the samples in order."""
    total = EvalResult()
    for (offset, length, codes, seqs, pad_flat, row_labels) in loss_batches:
        out = din.forward(codes, seqs, pad_flat)
        res = EvalResult(loss=bce_with_logits(out, row_labels) * length, count=length)      # Evaluator.scala:47-49
        for j in range(offset, offset + length):                                             # :51-66
            rec = tree.recommend_items(din, sequences[j], topk, candidate_num, use_mask=use_mask,
                                       consumed=user_consumed[int(users[j])])
            res.add_metrics(compute_metrics(rec, labels[j]))
        total = total + res
    return total


# ----------------------------------------------------------------------------------------------------------------------
# OTM evaluator restatement — O/ = /root/reference/otm/src/main/scala/com/mass/otm/
#   all_nodes        O/dataset/LocalDataSet.scala:199-205 (getAllNodes)
#   evaluate_otm     O/evaluation/Evaluator.scala:29-84 + computeLoss :86-96, Metrics.scala:7-31, EvalResult.scala:3-27;
#                    the search of one sample = CandidateSearcher.beamSearch through the C restatement (pyoracle.otm_beam_search)
# Parity status: unpinned (the reference's OtmModelTrainSpec only prints these numbers); follows the source line by line.

def all_nodes(ids):
    ids = [int(i) for i in ids]
    n = len(ids)
    leaf_level = 0
    while (1 << leaf_level) < n:                 # upperLog2
        leaf_level += 1
    res = set()
    for i in ids:
        a = i
        res.add(a)
        for _ in range(leaf_level):
            a = int((a - 1) / 2)                 # Scala Int division truncates toward zero
            res.add(a)
    return res


def evaluate_otm(search, sequences, labels, users, user_consumed, allowed, topk, total_eval_batch_size, beam_size, thread_num=1):
    """search(seq_codes) -> (node ids, f64 scores) of the leaf level in the reference's order."""
    n_all = len(sequences)
    batch = max(1, total_eval_batch_size // (beam_size * 2))
    total_loss, tp, tr, tn = 0.0, 0.0, 0.0, 0.0
    for off in range(0, n_all, batch):
        idx = list(range(off, min(n_all, off + batch)))
        tds = int(math.ceil(len(idx) / float(thread_num)))
        for c0 in range(0, len(idx), tds):
            preds, labs = [], []
            for j in idx[c0:c0 + tds]:
                ids, sc = search(sequences[j])
                consumed = set(int(x) for x in user_consumed[int(users[j])])
                nodes = [(int(i), float(s)) for i, s in zip(ids, sc) if int(i) not in consumed and int(i) in allowed]
                nodes = sorted(nodes, key=lambda t: -t[1])[:topk]          # Python's sort is stable, like sortBy
                tset = [int(t) for t in labels[j]]
                preds += [s for _, s in nodes]
                labs += [1.0 if i in tset else 0.0 for i, _ in nodes]
                m = compute_metrics([i for i, _ in nodes], tset)
                tp += m[0]; tr += m[1]; tn += m[2]
            a = b = 0.0
            for x, z in zip(preds, labs):
                a += max(x, 0.0) + math.log(math.exp(-abs(x)) + 1.0)
                b += x * z
            total_loss += a - b
    return total_loss / n_all, (tp / n_all, tr / n_all, tn / n_all)


# Deep-Retrieval evaluator restatement — D/ = /root/reference/deep-retrieval/src/main/scala/com/mass/dr/
#   evaluate_dr_metrics   D/evaluation/Evaluator.scala:41-70 (the metrics fold) + recommendItems :108-129, Metrics.scala:5-27
def evaluate_dr_metrics(dr, sequences, labels, users, user_consumed, topk, beam_size):
    p = r = g = 0.0
    for j in range(len(sequences)):
        consumed = set(int(x) for x in user_consumed[int(users[j])])
        paths, _ = dr.beam_search(sequences[j], beam_size)
        cands = [int(c) for c in dr.search_candidates(paths) if int(c) not in consumed]
        scores = dr.rerank(cands, sequences[j]) if cands else []
        order = sorted(range(len(cands)), key=lambda i: -scores[i])[:topk]
        m = compute_metrics([cands[i] for i in order], [int(t) for t in labels[j]])
        p += m[0]; r += m[1]; g += m[2]
    return p, r, g
