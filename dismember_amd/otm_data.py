"""OTM data set and mapping files: the host-side integer logic of
`com.mass.otm.dataset.LocalDataSet` (otm/src/main/scala/com/mass/otm/dataset/LocalDataSet.scala:19-206),
`TreeConstruction.readDataFile` (otm/.../tree/TreeConstruction.scala:366-416) and `Serialization.saveMapping / loadMapping`
(tdm/.../utils/Serialization.scala:103-125).  Everything downstream works in NODE-id space: an item is the leaf node its mapping
gives it, -1 is padding (otm/package.scala:9).

The reference draws the initial mapping and the epoch shuffles from an unseeded / JVM-specific `scala.util.Random`; here they come
from a seeded numpy generator — same distribution, not the same draw (parity for these steps is structural: leaf range, bijection,
sample counts)."""
import math

import numpy as np

PADDING = -1


def upper_log2(n):
    return int(math.ceil(math.log(n) / math.log(2)))          # otm/package.scala:16


def initialize_mapping(sample, leaf_init_mode="random", rng=None):
    """LocalDataSet.initializeMapping (:163-181) + sampleRandomLeaves (:183-191): distinct items in first-appearance order, ordered
    randomly or by (category, item), zipped with a sorted random subset of the leaves of level upperLog2(#items).
    -> dict item -> leaf node id"""
    rng = rng or np.random.default_rng(0)
    seen, uniq = set(), []
    for it, cat in zip(sample["item"], sample["category"]):
        if it not in seen:
            seen.add(it)
            uniq.append((int(it), cat))
    if leaf_init_mode == "random":
        items = [u[0] for u in uniq]
        items = [items[i] for i in rng.permutation(len(items))]
    elif leaf_init_mode == "category":
        items = [u[0] for u in sorted(uniq, key=lambda t: (t[1], t[0]))]
    else:
        raise ValueError("leaf_init_mode should either be `random` or `category`")
    n = len(items)
    leaf_level = upper_log2(n)
    start = (1 << leaf_level) - 1
    leaves = np.sort(start + rng.permutation(start + 1)[:n])
    return dict(zip(items, leaves.astype(int).tolist()))


def save_mapping(path, mapping):
    with open(path, "w") as f:
        for item, node in mapping.items():
            f.write("%d %d\n" % (item, node))


def load_mapping(path):
    out = {}
    with open(path) as f:
        for line in f:
            kv = line.split()
            if kv:
                out[int(kv[0])] = int(kv[-1])
    return out


def _user_nodes(sample, mapping):
    """per user: items sorted by time (stable), first occurrences, mapped to node ids; users in first-appearance order"""
    inter = {}
    for u, it, t in zip(sample["user"], sample["item"], sample["timestamp"]):
        inter.setdefault(int(u), []).append((int(it), t))
    out = {}
    for u, lst in inter.items():
        seen, seq = set(), []
        for it, _ in sorted(lst, key=lambda p: p[1]):
            if it not in seen:
                seen.add(it)
                seq.append(mapping[it])
        out[u] = seq
    return out


def _sliding(seq, size):
    """Scala's sliding(size): windows of `size` with step 1; a collection shorter than `size` yields itself once"""
    if len(seq) <= size:
        return [seq] if seq else []
    return [seq[i:i + size] for i in range(len(seq) - size + 1)]


def generate_samples(sample, mapping, seq_len, min_seq_len, split_ratio, label_num):
    """LocalDataSet.generateSamples (:71-104) -> (user_consumed {user: node ids}, train [(seq, labels, user)], eval [...])"""
    assert seq_len > 0 and min_seq_len > 0 and seq_len >= min_seq_len
    pad = [PADDING] * (seq_len - min_seq_len)
    consumed, train, evals = {}, [], []
    for user, items in _user_nodes(sample, mapping).items():
        if len(items) <= min_seq_len:
            continue
        if len(items) <= min_seq_len + label_num:
            train.append((pad + items[:min_seq_len], items[min_seq_len:], user))
            consumed[user] = list(items)
            continue
        full = pad + items
        split = int(math.ceil((len(items) - min_seq_len) * split_ratio))
        for s in _sliding(full[:split + seq_len], seq_len + label_num):
            train.append((s[:seq_len], s[seq_len:], user))
        consumed[user] = items[:split + min_seq_len]
        ev_seq, labels = full[:split + seq_len], full[split + seq_len:]
        evals.append((ev_seq[-seq_len:], labels, user))
    return consumed, train, evals


def item_sequences(sample, mapping, label_num, min_seq_len, seq_len, split_ratio):
    """TreeConstruction.readDataFile (:366-416): item -> the histories (node ids, flat [rows * seq_len]) of the training windows
    whose labels contain it — the itemSequenceMap of OTM tree construction."""
    back = {v: k for k, v in mapping.items()}
    pad = [PADDING] * (seq_len - min_seq_len)
    out = {}
    for _, items in _user_nodes(sample, mapping).items():
        if len(items) < min_seq_len + label_num:
            continue
        if len(items) == min_seq_len + label_num:
            full = pad + items[:min_seq_len]
            for lab in items[min_seq_len:]:
                out.setdefault(back[lab], []).extend(full)
            continue
        full = pad + items
        split = int(math.ceil((len(items) - min_seq_len) * split_ratio))
        for s in _sliding(full[:split + seq_len], seq_len + label_num):
            seq, labels = s[:seq_len], s[seq_len:]
            for lab in labels:
                out.setdefault(back[lab], []).extend(seq)
    return {k: np.asarray(v, np.int32) for k, v in out.items()}
