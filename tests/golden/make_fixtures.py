#!/usr/bin/env python3
"""Fixture generator (run in the BUILD container only; needs /root/reference).

Re-encodes the reference's bundled *data files* (BSD-3-Clause, see the
reference LICENSE) as plain little-endian numpy arrays so that the parity
tests can run on a box that has no /root/reference:

  data/jtm/example_tree.bin    -> tdm_tree.npz   (protobuf-KV tree, max_level 12; + the target statistics it was
                                  built from, and tdm_tree_file.json = size and sha256 of the original file)
  data/jtm/example_model.bin   -> din_f32.npy    (compact A0 weight vector, E=16, f32)
  data/otm/example_model.bin   -> din_f64.npy    (same layout, f64; OTM)
  data/otm/example_mapping.txt -> otm_mapping.npy (item -> leaf node id)
  data/example_data.csv        -> example_data.npz (interactions in time order + distinct items with category)
  data/dr/example_mapping.bin  -> dr_mapping.npz  (item, id, paths[J][D]; Deep-Retrieval)

Formats decoded here:
  * tree: `[int32 BE len][KVItem]` records, DistTree.loadData
    (tdm/src/main/scala/com/mass/tdm/tree/DistTree.scala:40-87) with the
    messages of tdm/src/main/protobuf/{store_kv,tree}.proto.
  * model: Java ObjectOutputStream; the first primitive array of length
    131857 is the compact parameter vector produced by Module.flatten
    (scalann/.../nn/mixin/Module.scala:9-45) in Graph.parameters order.

  * DR mapping: `[int32 BE size][ItemSet]`, MappingOp.loadMapping
    (deep-retrieval/src/main/scala/com/mass/dr/model/MappingOp.scala:71-95) with the messages of
    deep-retrieval/src/main/protobuf/item_mapping.proto (packed repeated int32 index).

Only DATA is re-encoded: no reference source travels.
"""
import os
import struct
import sys

import numpy as np

REF = os.environ.get("DM_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def varint(b, i):
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def pb_fields(b):
    """Minimal protobuf wire decoder -> list of (field, wiretype, value)."""
    i = 0
    out = []
    while i < len(b):
        tag, i = varint(b, i)
        f, w = tag >> 3, tag & 7
        if w == 0:
            v, i = varint(b, i)
        elif w == 2:
            n, i = varint(b, i)
            v = b[i:i + n]
            i += n
        elif w == 5:
            v = struct.unpack("<f", b[i:i + 4])[0]
            i += 4
        elif w == 1:
            v = struct.unpack("<d", b[i:i + 8])[0]
            i += 8
        else:
            raise ValueError("wire type %d" % w)
        out.append((f, w, v))
    return out


def to_i32(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def read_tree(path):
    data = open(path, "rb").read()
    i = 0
    nodes = {}
    pairs = []
    max_level = None
    while i < len(data):
        (n,) = struct.unpack(">i", data[i:i + 4])
        i += 4
        kv = dict((f, v) for f, _, v in pb_fields(data[i:i + n]))
        i += n
        key = kv[1].decode()
        val = kv.get(2, b"")
        if key.startswith("tree_meta"):
            for f, _, v in pb_fields(val):
                if f == 1:
                    max_level = v
        elif key.startswith("Part_"):
            for f, _, v in pb_fields(val):
                if f == 2:
                    d = dict((ff, vv) for ff, _, vv in pb_fields(v))
                    pairs.append((to_i32(d.get(1, 0)), to_i32(d.get(2, 0))))
        else:
            d = dict((ff, vv) for ff, _, vv in pb_fields(val))
            nodes[int(key)] = (to_i32(d.get(1, 0)), float(d.get(2, 0.0)), int(d.get(4, 0)))
    codes = np.array(sorted(nodes), dtype=np.int32)
    ids = np.array([nodes[c][0] for c in codes], dtype=np.int32)
    probs = np.array([nodes[c][1] for c in codes], dtype=np.float32)
    leaf = np.array([nodes[c][2] for c in codes], dtype=np.uint8)
    pairs = np.array(pairs, dtype=np.int32)
    return dict(codes=codes, ids=ids, probs=probs, is_leaf=leaf,
                leaf_ids=pairs[:, 0].copy(), leaf_codes=pairs[:, 1].copy(),
                max_level=np.int32(max_level))


def derive_stat(tree):
    """Item target counts (`stat`) that TreeBuilder.build was given, recovered from the node probabilities it wrote:
    a leaf's probability is its count (1.0 when the item had none), an ancestor's the sum over its leaves (1.0 when
    that sum is empty) — T/tree/TreeBuilder.scala:35-38,48-52,66-67,148-162.  Where 1.0 is ambiguous the parent's
    total decides; remaining ties produce identical files."""
    sys.setrecursionlimit(10000)
    prob = dict(zip(tree["codes"].tolist(), tree["probs"].tolist()))
    idof = dict(zip(tree["codes"].tolist(), tree["ids"].tolist()))
    leaf = dict(zip(tree["codes"].tolist(), tree["is_leaf"].tolist()))
    stat = {}

    def assign(node, total):
        if leaf[node]:
            if total > 0:
                stat[idof[node]] = int(total)
            return
        kids = [k for k in (2 * node + 1, 2 * node + 2) if k in prob]
        if len(kids) == 1:
            return assign(kids[0], total)
        a, b = kids
        amb = [prob[k] == 1.0 for k in kids]
        if not amb[0] and not amb[1]:
            ta, tb = prob[a], prob[b]
        elif amb[0] and not amb[1]:
            tb = prob[b]; ta = total - tb
        elif amb[1] and not amb[0]:
            ta = prob[a]; tb = total - ta
        else:
            ta = min(1, total); tb = total - ta
        assert ta >= 0 and tb >= 0 and ta + tb == total
        assign(a, ta); assign(b, tb)

    assign(0, prob[0])
    ids = np.array(sorted(stat), dtype=np.int32)
    return ids, np.array([stat[i] for i in ids], dtype=np.int32)


def read_weights(path, dtype, n=131857):
    b = open(path, "rb").read()
    i = b.find(struct.pack(">i", n))
    assert i >= 0
    return np.frombuffer(b, dtype=dtype, count=n, offset=i + 4).astype(dtype[1:])


def read_dr_mapping(path):
    data = open(path, "rb").read()
    (n,) = struct.unpack(">i", data[:4])
    items, ids, paths = [], [], []
    for f, _, v in pb_fields(data[4:4 + n]):
        assert f == 1
        item = idx = 0
        ps = []
        for ff, ww, vv in pb_fields(v):
            if ff == 1:
                item = to_i32(vv)
            elif ff == 2:
                idx = to_i32(vv)
            elif ff == 3:
                p = []
                for f3, w3, v3 in pb_fields(vv):
                    if w3 == 2:      # packed
                        i = 0
                        while i < len(v3):
                            x, i = varint(v3, i)
                            p.append(x)
                    else:
                        p.append(v3)
                ps.append(p)
        items.append(item)
        ids.append(idx)
        paths.append(ps)
    return dict(items=np.array(items, np.int32), ids=np.array(ids, np.int32), paths=np.array(paths, np.int32))


def main():
    tree = read_tree(os.path.join(REF, "data/jtm/example_tree.bin"))
    assert tree["max_level"] == 12 and len(tree["leaf_ids"]) == 3706
    tree["stat_ids"], tree["stat_counts"] = derive_stat(tree)
    np.savez_compressed(os.path.join(OUT, "tdm_tree.npz"), **tree)
    import hashlib
    import json
    raw = open(os.path.join(REF, "data/jtm/example_tree.bin"), "rb").read()
    json.dump({"file": "data/jtm/example_tree.bin", "bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest()},
              open(os.path.join(OUT, "tdm_tree_file.json"), "w"))
    w32 = read_weights(os.path.join(REF, "data/jtm/example_model.bin"), ">f4")
    assert abs(float(w32[-1]) - (-0.1667024)) < 1e-7
    np.save(os.path.join(OUT, "din_f32.npy"), w32.astype("<f4"))
    w64 = read_weights(os.path.join(REF, "data/otm/example_model.bin"), ">f8")
    assert abs(float(w64[-1]) - (-0.22942817)) < 1e-8
    np.save(os.path.join(OUT, "din_f64.npy"), w64.astype("<f8"))
    m = np.loadtxt(os.path.join(REF, "data/otm/example_mapping.txt"), dtype=np.int64)
    np.save(os.path.join(OUT, "otm_mapping.npy"), m.astype(np.int32))
    # data/example_data.csv (user,item,label,timestamp,genre; 100 000 MovieLens-1M interactions): rows in stable
    # timestamp order (timestamps themselves dropped: only their order matters to TreeInit.getUserInteracted) and
    # the distinct (item, category index) pairs in file order (TreeInit.initializeTree / readFile).
    us, its, ts, cats, cd = [], [], [], [], {}
    for line in open(os.path.join(REF, "data/example_data.csv")):
        arr = line.strip().split(",")
        if len(arr) != 5 or not arr[0].isdigit():
            continue
        us.append(int(arr[0])); its.append(int(arr[1])); ts.append(int(arr[3]))
        cats.append(cd.setdefault(arr[4], len(cd)))
    order = np.argsort(np.array(ts), kind="stable")
    seen, ui, uc = set(), [], []
    for it, c in zip(its, cats):
        if it not in seen:
            seen.add(it); ui.append(it); uc.append(c)
    np.savez_compressed(os.path.join(OUT, "example_data.npz"), user=np.array(us, np.int16)[order],
                        item=np.array(its, np.int16)[order], uniq_item=np.array(ui, np.int16), uniq_cat=np.array(uc, np.int8))
    dr = read_dr_mapping(os.path.join(REF, "data/dr/example_mapping.bin"))
    assert dr["paths"].shape == (3325, 2, 3) and dr["paths"].max() < 100
    np.savez_compressed(os.path.join(OUT, "dr_mapping.npz"), **dr)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    sys.exit(main())
