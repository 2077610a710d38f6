#!/usr/bin/env python3
"""Fixture generator (run in the BUILD container only; needs /root/reference).

Re-encodes the reference's bundled *data files* (BSD-3-Clause, see the
reference LICENSE) as plain little-endian numpy arrays so that the parity
tests can run on a box that has no /root/reference:

  data/jtm/example_tree.bin    -> tdm_tree.npz   (protobuf-KV tree, max_level 12)
  data/jtm/example_model.bin   -> din_f32.npy    (compact A0 weight vector, E=16, f32)
  data/otm/example_model.bin   -> din_f64.npy    (same layout, f64; OTM)
  data/otm/example_mapping.txt -> otm_mapping.npy (item -> leaf node id)
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
    np.savez_compressed(os.path.join(OUT, "tdm_tree.npz"), **tree)
    w32 = read_weights(os.path.join(REF, "data/jtm/example_model.bin"), ">f4")
    assert abs(float(w32[-1]) - (-0.1667024)) < 1e-7
    np.save(os.path.join(OUT, "din_f32.npy"), w32.astype("<f4"))
    w64 = read_weights(os.path.join(REF, "data/otm/example_model.bin"), ">f8")
    assert abs(float(w64[-1]) - (-0.22942817)) < 1e-8
    np.save(os.path.join(OUT, "din_f64.npy"), w64.astype("<f8"))
    m = np.loadtxt(os.path.join(REF, "data/otm/example_mapping.txt"), dtype=np.int64)
    np.save(os.path.join(OUT, "otm_mapping.npy"), m.astype(np.int32))
    dr = read_dr_mapping(os.path.join(REF, "data/dr/example_mapping.bin"))
    assert dr["paths"].shape == (3325, 2, 3) and dr["paths"].max() < 100
    np.savez_compressed(os.path.join(OUT, "dr_mapping.npz"), **dr)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    sys.exit(main())
