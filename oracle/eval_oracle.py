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
