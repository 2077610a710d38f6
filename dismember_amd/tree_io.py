"""Tree files and tree initialisation — host-side integer logic around the device index.

T/ = tdm/src/main/scala/com/mass/tdm/ of the reference.

  read_tree_file     DistTree.loadData                 T/tree/DistTree.scala:40-87   ([int32 BE len][KVItem] records)
  build_tree_bytes   TreeBuilder.build                 T/tree/TreeBuilder.scala:23-101 (+ flattenLeaves :133-141,
                     getAncestors :143-146, computeNodeOccurrence :148-162); byte-identical files
  gen_codes          TreeInit.initializeTree / genCode T/tree/TreeInit.scala:180-226
  read_interactions  TreeInit.readFile                 T/tree/TreeInit.scala:50-97
  user_sequences     TreeInit.getUserInteracted        T/tree/TreeInit.scala:99-118
  split_samples      TreeInit.writeTrain / writeEither T/tree/TreeInit.scala:242-333 (sliding windows, split ratio,
                     user_consumed, target statistics)
Messages: tdm/src/main/protobuf/{store_kv,tree}.proto, encoded as ScalaPB does (fields in number order, proto3
defaults omitted).
"""
import math
import struct

import numpy as np


# ------------------------------------------------------------------ protobuf wire helpers
def _varint(v):
    v &= (1 << 64) - 1            # int32 negatives are sign-extended to 10 bytes, like protobuf
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _f_varint(field, v):
    return b"" if v == 0 else _varint(field << 3) + _varint(int(v))


def _f_bytes(field, b, keep_empty=False):
    return b"" if (not b and not keep_empty) else _varint((field << 3) | 2) + _varint(len(b)) + b


def _f_float(field, x):
    bits = struct.pack("<f", x)
    return b"" if bits == b"\x00\x00\x00\x00" else _varint((field << 3) | 5) + bits


def _node(node_id, prob, is_leaf):
    # Node(id, probality, leaf_cate_id = 0, is_leaf)   tree.proto
    return _f_varint(1, node_id) + _f_float(2, prob) + _f_varint(4, 1 if is_leaf else 0)


def _kv(key, value):
    msg = _f_bytes(1, key) + _f_bytes(2, value)
    return struct.pack(">i", len(msg)) + msg          # TreeBuilder.writeKV: 4-byte big-endian size, then the message


def _read_varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _fields(b):
    i, out = 0, []
    while i < len(b):
        tag, i = _read_varint(b, i)
        f, w = tag >> 3, tag & 7
        if w == 0:
            v, i = _read_varint(b, i)
        elif w == 2:
            n, i = _read_varint(b, i)
            v = b[i:i + n]
            i += n
        elif w == 5:
            v = struct.unpack("<f", b[i:i + 4])[0]
            i += 4
        elif w == 1:
            v = struct.unpack("<d", b[i:i + 8])[0]
            i += 8
        else:
            raise ValueError("unsupported wire type %d" % w)
        out.append((f, v))
    return out


def _i32(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


# ------------------------------------------------------------------ reader
def read_tree_bytes(data):
    """-> dict(codes, ids, probs, is_leaf, leaf_ids, leaf_codes, max_level): the arrays Engine.load_tree /
    load_id_maps take (DistTree.loadData keeps codeNodeMap and idCodeMap, DistTree.scala:40-87)."""
    i = 0
    nodes, pairs, max_level = {}, [], None
    while i < len(data):
        (n,) = struct.unpack(">i", data[i:i + 4])
        i += 4
        kv = dict(_fields(data[i:i + n]))
        i += n
        key = kv[1].decode()
        val = kv.get(2, b"")
        if key.startswith("tree_meta"):
            for f, v in _fields(val):
                if f == 1:
                    max_level = v
        elif key.startswith("Part_"):
            for f, v in _fields(val):
                if f == 2:
                    d = dict(_fields(v))
                    pairs.append((_i32(d.get(1, 0)), _i32(d.get(2, 0))))
        else:
            d = dict(_fields(val))
            nodes[int(key)] = (_i32(d.get(1, 0)), float(d.get(2, 0.0)), int(d.get(4, 0)))
    if max_level is None:
        raise ValueError("tree file has no tree_meta record")
    codes = np.array(sorted(nodes), dtype=np.int32)
    pairs = np.array(pairs, dtype=np.int32).reshape(-1, 2)
    return dict(codes=codes, ids=np.array([nodes[c][0] for c in codes], np.int32),
                probs=np.array([nodes[c][1] for c in codes], np.float32),
                is_leaf=np.array([nodes[c][2] for c in codes], np.uint8),
                leaf_ids=pairs[:, 0].copy(), leaf_codes=pairs[:, 1].copy(), max_level=int(max_level))


def read_max_level(data):
    """TreeMeta.max_level of a tree file's bytes (the tree_meta record), without materialising the nodes."""
    i = 0
    while i < len(data):
        (n,) = struct.unpack(">i", data[i:i + 4])
        i += 4
        if b"tree_meta" in data[i:i + min(n, 24)]:
            kv = dict(_fields(data[i:i + n]))
            for f, v in _fields(kv.get(2, b"")):
                if f == 1:
                    return int(v)
        i += n
    raise ValueError("tree file has no tree_meta record")


def read_tree_file(path):
    with open(path, "rb") as f:
        return read_tree_bytes(f.read())


# ------------------------------------------------------------------ writer
def flatten_leaves(codes, min_code):
    out = []
    for c in codes:                                   # sink(): code*2+1 until the last level
        c = int(c)
        while c < min_code:
            c = c * 2 + 1
        out.append(c)
    return out


def get_ancestors(code, max_level):
    out = []
    for _ in range(max_level):
        code = (code - 1) // 2 if code > 0 else 0      # Scala's (a - 1) / 2 truncates toward zero: (0 - 1) / 2 = 0
        out.append(code)
    return out


def build_tree_bytes(tree_ids, tree_codes, stat=None):
    """TreeBuilder.build: the file bytes.  stat: dict item id -> count (None = all probabilities 1.0)."""
    tree_ids = [int(x) for x in tree_ids]
    tree_codes = [int(x) for x in tree_codes]
    offset = max(0, max(tree_ids)) + 1
    max_level = int(math.floor(math.log(max(tree_codes) + 1) / math.log(2)))
    min_leaf = int(math.pow(2, max_level)) - 1
    leaf_codes = flatten_leaves(tree_codes, min_leaf)
    items = sorted(zip(tree_ids, leaf_codes), key=lambda t: t[1])        # stable sortBy(_.code)
    pstat = {}
    if stat is not None:                                                  # computeNodeOccurrence (float32 sums)
        for item, code in items:
            if item in stat:
                for anc in get_ancestors(code, max_level):
                    pstat[anc] = np.float32(pstat.get(anc, np.float32(0.0)) + np.float32(stat[item]))
    out = bytearray()
    parts, tmp, saved = [], [], set()
    for i, (item, code) in enumerate(items):
        prob = float(stat[item]) if (stat is not None and item in stat) else 1.0
        out += _kv(str(code).encode(), _node(item, prob, True))
        tmp.append((item, code))
        if i == len(items) - 1 or len(tmp) == 512:
            pid = ("Part_%d" % (len(parts) + 1)).encode()
            body = _f_bytes(1, pid) + b"".join(_f_bytes(2, _f_varint(1, a) + _f_varint(2, b), keep_empty=True) for a, b in tmp)
            parts.append((pid, body))
            tmp = []
        for anc in get_ancestors(code, max_level):
            if anc not in saved:
                out += _kv(str(anc).encode(), _node(anc + offset, float(pstat.get(anc, 1.0)), False))
                saved.add(anc)
    for pid, body in parts:
        out += _kv(pid, body)
    meta = _f_varint(1, max_level) + b"".join(_f_bytes(2, pid, keep_empty=True) for pid, _ in parts)
    out += _kv(b"tree_meta", meta)
    return bytes(out)


def build_jtm_tree_bytes(item_ids, new_codes, leaf_probs, max_level, non_leaf_offset):
    """JTMTree.writeTree (jtm/src/main/scala/com/mass/jtm/tree/JTMTree.scala:115-182): the tree file of a learned
    projection pi (item -> new leaf code).  leaf_probs[i] = the probability item i's OLD leaf node carried.  Every
    ancestor's probability is the float32 sum over its leaves.  Records follow the order of the given items (the
    reference iterates a Scala Map, i.e. hash order; readers are order-independent)."""
    leaf_stat, pstat = {}, {}
    for it, code, pr in zip(item_ids, new_codes, leaf_probs):
        code = int(code)
        leaf_stat[code] = np.float32(pr)
        for anc in get_ancestors(code, max_level):
            pstat[anc] = np.float32(pstat.get(anc, np.float32(0.0)) + np.float32(pr))
    out = bytearray()
    parts, tmp, saved = [], [], set()
    n = len(item_ids)
    for i, (it, code) in enumerate(zip(item_ids, new_codes)):
        it, code = int(it), int(code)
        out += _kv(str(code).encode(), _node(it, float(leaf_stat[code]), True))
        tmp.append((it, code))
        if i == n - 1 or len(tmp) == 512:
            pid = ("Part_%d" % (len(parts) + 1)).encode()
            body = _f_bytes(1, pid) + b"".join(_f_bytes(2, _f_varint(1, a) + _f_varint(2, b), keep_empty=True) for a, b in tmp)
            parts.append((pid, body))
            tmp = []
        for anc in get_ancestors(code, max_level):
            if anc not in saved:
                out += _kv(str(anc).encode(), _node(anc + non_leaf_offset, float(pstat[anc]), False))
                saved.add(anc)
    for pid, body in parts:
        out += _kv(pid, body)
    meta = _f_varint(1, max_level) + b"".join(_f_bytes(2, pid, keep_empty=True) for pid, _ in parts)
    out += _kv(b"tree_meta", meta)
    return bytes(out)


def write_tree_file(path, tree_ids, tree_codes, stat=None):
    with open(path, "wb") as f:
        f.write(build_tree_bytes(tree_ids, tree_codes, stat))


# ------------------------------------------------------------------ tree initialisation
def gen_codes(item_ids, cat_ids):
    """initializeTree: unique items in first-appearance order -> (ids, codes) sorted by (category, id), codes from the
    recursive halving genCode (upper half gets child 2c+1, lower half 2c+2)."""
    seen, uniq = set(), []
    for it, c in zip(item_ids, cat_ids):
        it = int(it)
        if it not in seen:
            seen.add(it)
            uniq.append((it, int(c)))
    items = sorted(uniq, key=lambda t: (t[1], t[0]))
    codes = [0] * len(items)
    stack = [(0, len(items), 0)]
    while stack:
        start, end, code = stack.pop()
        if end <= start:
            continue
        if end == start + 1:
            codes[start] = code
            continue
        mid = (start + end) >> 1
        stack.append((mid, end, 2 * code + 1))
        stack.append((start, mid, 2 * code + 2))
    return [t[0] for t in items], codes, [t[0] for t in uniq]


def _is_creatable(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def read_interactions(lines):
    """TreeInit.readFile over `user,item,label,timestamp,category` lines (header and malformed lines skipped)."""
    cat_dict = {}
    users, items, cats, times = [], [], [], []
    for line in lines:
        arr = line.strip().split(",")
        if len(arr) != 5 or not _is_creatable(arr[0]):
            continue
        users.append(int(arr[0])); items.append(int(arr[1])); times.append(int(arr[3]))
        if arr[4] not in cat_dict:
            cat_dict[arr[4]] = len(cat_dict)
        cats.append(cat_dict[arr[4]])
    return dict(user=users, item=items, category=cats, timestamp=times)


def user_sequences(sample):
    """getUserInteracted: per user, items sorted by time (stable), first occurrences only.  Users in first-appearance
    order (the reference iterates a HashMap; the order only affects the line order of the files it writes)."""
    inter = {}
    for u, it, t in zip(sample["user"], sample["item"], sample["timestamp"]):
        inter.setdefault(u, []).append((it, t))
    out = {}
    for u, lst in inter.items():
        seen, seq = set(), []
        for it, _ in sorted(lst, key=lambda p: p[1]):
            if it not in seen:
                seen.add(it)
                seq.append(it)
        out[u] = seq
    return out


def split_samples(user_items, seq_len, min_seq_len, split_for_eval=True, split_ratio=0.8):
    """writeTrain / writeEither: -> dict(train=[(name, [seq_len+1 ids])], eval=[(name, seq [seq_len], labels)],
    user_consumed={user: ids}, stat={item: target count})."""
    assert seq_len > 0 and min_seq_len > 0 and seq_len >= min_seq_len and 0 < split_ratio < 1
    train, evals, consumed, stat = [], [], {}, {}
    pad = [0] * (seq_len - min_seq_len)
    if not split_for_eval:
        for user, items in user_items.items():
            consumed[user] = list(items)
            if len(items) > min_seq_len:
                arr = pad + list(items)
                for ui in range(len(arr) - seq_len):
                    seq = arr[ui:ui + seq_len + 1]
                    train.append(("%d_%d" % (user, ui), seq))
                    stat[seq[-1]] = stat.get(seq[-1], 0) + 1
        return dict(train=train, eval=evals, user_consumed=consumed, stat=stat)
    for user, items in user_items.items():              # train half
        items = list(items)
        if len(items) <= min_seq_len:
            consumed[user] = items
            continue
        arr = pad + items
        train_num = int(math.ceil((len(items) - min_seq_len) * split_ratio))
        consumed[user] = items if len(items) == min_seq_len + 1 else items[:train_num + min_seq_len]
        for i in range(train_num):
            seq = arr[i:i + seq_len + 1]
            train.append(("user_%d_%d" % (user, i), seq))
            stat[seq[-1]] = stat.get(seq[-1], 0) + 1
    for user, items in user_items.items():              # eval half
        items = list(items)
        if len(items) <= min_seq_len + 1:
            continue
        arr = pad + items
        split_point = int(math.ceil((len(items) - min_seq_len) * split_ratio))
        cons = set(consumed[user])
        seq = arr[split_point:split_point + seq_len]
        labels = [x for x in arr[split_point + seq_len:] if x not in cons]   # drop items that appeared in train data
        if labels:
            evals.append(("user_%d" % user, seq, labels))
    return dict(train=train, eval=evals, user_consumed=consumed, stat=stat)


def initialize_tree(lines, seq_len=10, min_seq_len=2, split_for_eval=True, split_ratio=0.8):
    """TreeInit.generate without the file outputs: -> (tree file bytes, ids, codes, split)."""
    sample = read_interactions(lines)
    split = split_samples(user_sequences(sample), seq_len, min_seq_len, split_for_eval, split_ratio)
    ids, codes, _ = gen_codes(sample["item"], sample["category"])
    return build_tree_bytes(ids, codes, split["stat"]), ids, codes, split


# ------------------------------------------------------------------ data files (TreeInit writers / LocalDataSet readers)
def write_split_files(split, train_path, eval_path=None, stat_path=None, user_consumed_path=None):
    """The text files TreeInit.generate writes (T/tree/TreeInit.scala:120-176,333-360)."""
    with open(train_path, "w") as f:
        for name, seq in split["train"]:
            f.write(name + "," + ",".join(str(x) for x in seq) + "\n")
    if eval_path is not None:
        with open(eval_path, "w") as f:
            for name, seq, labels in split["eval"]:
                f.write(name + "," + ",".join(str(x) for x in list(seq) + list(labels)) + "\n")
    if stat_path is not None:
        with open(stat_path, "w") as f:
            for item, cnt in split["stat"].items():
                f.write("%d, %d\n" % (item, cnt))
    if user_consumed_path is not None:
        with open(user_consumed_path, "w") as f:
            for user, items in split["user_consumed"].items():
                f.write("user_%d" % user + "".join(",%d" % i for i in items) + "\n")


def read_train_data(lines):
    """LocalDataSet.readTrainData (T/dataset/LocalDataSet.scala:148-158): `name,seq...,target`; all-padding
    sequences are dropped.  -> (sequences [N, L] int32, targets [N] int32)"""
    seqs, tgts = [], []
    for line in lines:
        arr = line.strip().split(",")
        if len(arr) < 3:
            continue
        seq = [int(float(x)) for x in arr[1:-1]]
        if any(v != 0 for v in seq):
            seqs.append(seq)
            tgts.append(int(arr[-1]))
    return np.array(seqs, np.int32), np.array(tgts, np.int32)


def read_eval_data(lines, seq_len):
    """LocalDataSet.readEvalData (:160-171): `user_<id>,seq (seq_len),labels...` -> (sequences, labels list, users)"""
    seqs, labels, users = [], [], []
    for line in lines:
        arr = line.strip().split(",")
        if len(arr) < seq_len + 2:
            continue
        users.append(int(arr[0][5:]))
        seqs.append([int(x) for x in arr[1:seq_len + 1]])
        labels.append(np.array([int(x) for x in arr[seq_len + 1:]], np.int32))
    return np.array(seqs, np.int32), labels, np.array(users, np.int64)


def read_user_consumed(lines):
    """LocalDataSet.readUserConsumed (:173-182): `user_<id>,items...` -> dict"""
    out = {}
    for line in lines:
        arr = line.strip().split(",")
        if arr and arr[0].startswith("user_"):
            out[int(arr[0][5:])] = np.array([int(x) for x in arr[1:]], np.int32)
    return out
