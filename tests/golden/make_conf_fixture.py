#!/usr/bin/env python3
"""Fixture generator (BUILD container only; needs /root/reference): the KEY LISTS of the reference's configs/*.conf per
prefix, read with the product's own reader (dismember_amd/conf.py), and the typed values of the keys the hot path consumes.
Output: tests/golden/conf_keys.json.  No reference file travels: the fixture holds key names and a handful of scalar values."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = os.environ.get("DM_REFERENCE", "/root/reference")

from dismember_amd import conf  # noqa: E402

FILES = {"tdm": ["init", "model", "cluster"], "jtm": ["init", "model", "tree"], "otm": ["model", "tree"],
         "deep-retrieval": ["model", "cd"]}
HOT = ["seq_len", "embed_size", "beam_size", "topk_number", "layer_negative_counts", "start_sample_level",
       "sample_with_probability", "sample_tolerance", "learning_rate", "total_batch_size", "thread_number", "gap", "label_num",
       "num_layer", "num_node", "num_path_per_item", "candidate_path_num", "deep_model"]


def main():
    out = {}
    for name, prefixes in FILES.items():
        path = os.path.join(REF, "configs", name + ".conf")
        out[name] = {}
        for p in prefixes:
            c = conf.read_conf(path, p)
            out[name][p] = {"keys": sorted(c), "hot_values": {k: c[k] for k in HOT if k in c}}
    json.dump(out, open(os.path.join(HERE, "conf_keys.json"), "w"), indent=1, sort_keys=True)
    print("wrote conf_keys.json:", {n: {p: len(v["keys"]) for p, v in d.items()} for n, d in out.items()})


if __name__ == "__main__":
    main()
