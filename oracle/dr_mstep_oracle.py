"""TEST INFRASTRUCTURE — restatement of the Deep-Retrieval M-step, pure Python loops.
D/ = /root/reference/deep-retrieval/src/main/scala/com/mass/dr/.

  batch_path_score      CoordinateDescent.batchPathScore + aggregatePathScore  D/optim/CoordinateDescent.scala:117-163
  streaming_path_score  CoordinateDescent.streamingPathScore                    D/optim/CoordinateDescent.scala:165-211
  optimize              CoordinateDescent.optimize                              D/optim/CoordinateDescent.scala:29-83
  penalty_func          D/optim/CoordinateDescent.scala:112-115
Where the reference's order comes from a hash map / hash set (groupMapReduce(...).toSeq, keySet.union, idItemMapping.keys)
this file uses a defined order instead — ascending path tuple, ascending item id — and says so at the spot; those orders
only decide ties (and, for the item loop, which item sees a path's penalty first).  Parity status: unpinned at the JVM
boundary (no reference test pins an assignment; CoordinateDescentSpec needs the absent data/dr/example_model.bin);
pinned on hand-computed known answers in tests/test_dr_mstep.py.
"""
import math


def penalty_func(path_size, poly_order):
    f = lambda s: math.pow(s, poly_order) / poly_order
    return f(path_size + 1) - f(path_size)


def _sorted_desc(scores):
    """sortBy(_.prob)(Ordering[Double].reverse): stable, on a sequence in ascending path order (see the header)."""
    return sorted(sorted(scores, key=lambda t: t[0]), key=lambda t: -t[1])


def batch_path_score(samples, beam_search, num_candidate_path):
    """samples: [(sequence, target)]; beam_search(sequence, beam) -> [(path tuple, prob)].  -> {item: [(path, prob)]}"""
    per_item = {}
    for seq, target in samples:
        per_item.setdefault(target, []).append(beam_search(seq, num_candidate_path))
    out = {}
    for item, lists in per_item.items():
        acc = {}
        for lst in lists:                      # groupMapReduce(_.path)(_.prob)(_ + _): sums in sample order
            for path, prob in lst:
                acc[path] = acc[path] + prob if path in acc else prob
        out[item] = _sorted_desc(list(acc.items()))[:num_candidate_path]
    return out


def streaming_path_score(samples, beam_search, num_candidate_path, decay_factor, batch_size):
    scores = {}
    for b in range(0, len(samples), batch_size):
        for seq, item in samples[b:b + batch_size]:
            cand = beam_search(seq, num_candidate_path)
            if item not in scores:
                scores[item] = list(cand)
                continue
            orig = scores[item]
            min_score = min(p for _, p in orig)
            o, c = dict(orig), dict(cand)
            new = []
            for path in sorted(set(o) | set(c)):           # keySet.union(...).toSeq: hash order in the reference
                if path in o and path in c:
                    s = decay_factor * o[path] + c[path]
                elif path in c:
                    s = decay_factor * min_score + c[path]
                else:
                    s = decay_factor * o[path]
                new.append((path, s))
            scores[item] = sorted(new, key=lambda t: -t[1])[:num_candidate_path]
    return scores


def optimize(item_path_score, item_occurrence, all_items, num_iteration, num_path_per_item, random_paths,
             penalty_factor=3e-6, penalty_poly_order=4):
    """all_items: iteration order of `dataset.idItemMapping.keys` (ascending id here); random_paths(item) -> J paths for
    items that never occur as a target (the reference draws them from an unseeded Random)."""
    mapping, path_size = {}, {}
    for t in range(1, num_iteration + 1):
        for v in all_items:
            if v not in item_occurrence:
                mapping[v] = random_paths(v)
                continue
            selected, partial = [], 0.0
            for j in range(num_path_per_item - 1, -1, -1):          # List.range(0, J).foldRight
                if t > 1:
                    last = mapping[v][j]
                    path_size[last] = path_size[last] - 1
                cand = [n for n in item_path_score[v] if n[0] not in selected] if selected else item_path_score[v]
                best, best_score = None, None
                for path, prob in cand:
                    size = path_size.get(path, 0)
                    penalty = penalty_factor * penalty_func(size, penalty_poly_order)
                    nv = item_occurrence[v]
                    g = nv * (math.log1p(prob + partial) - math.log1p(partial)) - penalty
                    if best is None or g > best_score:              # maxBy: the first maximum
                        best, best_score = path, g
                if best is None:
                    raise ValueError("empty.maxBy")                 # fewer candidate paths than paths per item
                path_size[best] = path_size.get(best, 0) + 1
                selected = [best] + selected
                partial = partial + best_score                      # sic: the GAIN, not the probability, is accumulated
            mapping[v] = selected
    return mapping
