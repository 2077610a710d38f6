"""`.conf` reader and task table: the reference's configuration surface for this path.

Property.readConf / getOrStop / configLocal (scalann/src/main/scala/com/mass/scalann/utils/Property.scala:12-71) and the
ten CommandApp mains of examples/src/main/scala/com/mass/retrieval/{tdm,jtm,otm,dr}/ with their getParameters tables
(`tdm/package.scala:59-111`, `jtm/package.scala:20-36`, `otm/package.scala:48-93`, `dr/package.scala:52-99`).
The file format is kept verbatim — lines `prefix.key<whitespace>value`, filtered by prefix, split on whitespace, exactly two
tokens, later duplicates win (`Map(lines: _*)`) — so the reference's configs/*.conf load unchanged; key names, required /
optional status and defaults are the reference's (doc/configuration.md).
"""
import os
import re

_WS = " \t\n\x0b\f\r"          # java.util.regex \s
_SPLIT = re.compile("[" + re.escape(_WS) + "]+")


def _java_trim(s):
    """String.trim: strip every char <= U+0020 from both ends."""
    i, j = 0, len(s)
    while i < j and s[i] <= " ":
        i += 1
    while j > i and s[j - 1] <= " ":
        j -= 1
    return s[i:j]


def read_conf(path, prefix, truncate=True):
    """Property.readConf (Property.scala:12-49): {key: value} of the lines that START with `prefix`."""
    if not os.path.exists(path):
        raise ValueError("requirement failed: Config file %s doesn't exist" % path)       # `require`, :25
    out = {}
    with open(path, "r", encoding="utf-8") as f:
        for line in f.read().splitlines():
            if not line.startswith(prefix):                # :31 (a leading blank or '#' drops the line)
                continue
            tok = _SPLIT.split(_java_trim(line))           # :32
            if len(tok) != 2:                              # :33 (a value with blanks is silently dropped, as in the reference)
                continue
            key = tok[0][len(prefix) + 1:] if truncate else tok[0]       # :35
            out[key] = tok[1]                              # Map(lines: _*): the last occurrence wins
    return out


def get_or_stop(conf, key):
    """Property.getOrStop (:66-71)."""
    if key not in conf:
        raise ValueError("failed to read parameter: %s in conf file" % key)
    return conf[key]


def core_number(conf_num):
    """Property.getCoreNumber (:57-64): thread_number <= 0 means every available processor."""
    conf_num = int(conf_num)
    return (os.cpu_count() or 1) if conf_num <= 0 else conf_num


def _bool(s):
    # scala StringOps.toBoolean: "true" / "false", case-insensitive, anything else throws
    t = s.lower()
    if t not in ("true", "false"):
        raise ValueError('For input string: "%s"' % s)
    return t == "true"


REQ = object()      # marker: getOrStop
OPT = None          # marker: conf.get -> Option

# task name (the CommandApp object) -> (command-line flag, prefix, resource name, [(conf key, type, default | REQ | OPT)])
TASKS = {
    "TDMInitializeTree": ("tdmConfFile", "init", "tdm", [
        ("seq_len", int, REQ), ("min_seq_len", int, REQ), ("split_for_eval", _bool, "true"), ("split_ratio", float, "0.8"),
        ("data_path", str, REQ), ("train_path", str, REQ), ("eval_path", str, OPT), ("stat_path", str, REQ),
        ("leaf_id_path", str, REQ), ("tree_protobuf_path", str, REQ), ("user_consumed_path", str, OPT)]),
    "TDMTrainDeepModel": ("tdmConfFile", "model", "tdm", [
        ("deep_model", str.lower, REQ), ("seq_len", int, REQ), ("total_batch_size", int, REQ), ("total_eval_batch_size", int, REQ),
        ("layer_negative_counts", str, REQ), ("sample_with_probability", _bool, "true"), ("start_sample_level", int, "1"),
        ("sample_tolerance", int, "20"), ("thread_number", int, REQ), ("train_path", str, REQ), ("eval_path", str, REQ),
        ("tree_protobuf_path", str, REQ), ("user_consumed_path", str, REQ), ("embed_size", int, REQ), ("learning_rate", float, REQ),
        ("iteration_number", int, "100"), ("show_progress_interval", int, "1"), ("topk_number", int, "10"), ("beam_size", int, "20"),
        ("model_path", str, REQ), ("embed_path", str, REQ)]),
    "TDMClusterTree": ("tdmConfFile", "cluster", "tdm", [
        ("embed_path", str, REQ), ("parallel", _bool, "true"), ("thread_number", int, REQ), ("cluster_num", int, "10"),
        ("tree_protobuf_path", str, REQ), ("cluster_type", str, "kmeans")]),
    "JTMTreeLearning": ("jtmConfFile", "tree", "jtm", [
        ("deep_model", str.lower, REQ), ("data_path", str, REQ), ("tree_protobuf_path", str, REQ), ("model_path", str, REQ),
        ("gap", int, REQ), ("seq_len", int, REQ), ("hierarchical_preference", _bool, REQ), ("min_level", int, REQ),
        ("thread_number", int, REQ)]),
    "OTMTrainDeepModel": ("otmConfFile", "model", "otm", [
        ("deep_model", str.lower, REQ), ("data_path", str, REQ), ("model_path", str, REQ), ("thread_number", int, REQ),
        ("train_batch_size", int, REQ), ("eval_batch_size", int, REQ), ("embed_size", int, REQ), ("learning_rate", float, REQ),
        ("epoch_num", int, REQ), ("topk_number", int, REQ), ("beam_size", int, REQ), ("show_progress_interval", int, REQ),
        ("seq_len", int, REQ), ("min_seq_len", int, REQ), ("split_ratio", float, REQ), ("leaf_init_mode", str, REQ),
        ("initialize_mapping", _bool, REQ), ("mapping_path", str, REQ), ("label_num", int, REQ), ("target_mode", str, REQ),
        ("seed", int, REQ)]),
    "OTMConstructTree": ("otmConfFile", "tree", "otm", [
        ("deep_model", str.lower, REQ), ("data_path", str, REQ), ("model_path", str, REQ), ("mapping_path", str, REQ),
        ("thread_number", int, REQ), ("gap", int, REQ), ("label_num", int, REQ), ("seq_len", int, REQ), ("min_seq_len", int, REQ),
        ("split_ratio", float, REQ)]),
    "DRTrainDeepModel": ("drConfFile", "model", "deep-retrieval", [
        ("data_path", str, REQ), ("model_path", str, REQ), ("mapping_path", str, REQ), ("thread_number", int, REQ),
        ("train_batch_size", int, REQ), ("eval_batch_size", int, REQ), ("num_layer", int, REQ), ("num_node", int, REQ),
        ("num_path_per_item", int, REQ), ("embed_size", int, REQ), ("learning_rate", float, REQ), ("epoch_num", int, REQ),
        ("num_sampled", int, REQ), ("topk_number", int, REQ), ("beam_size", int, REQ), ("show_progress_interval", int, REQ),
        ("seq_len", int, REQ), ("min_seq_len", int, REQ), ("split_ratio", float, REQ), ("initialize_mapping", _bool, REQ)]),
    "DRCoordinateDescent": ("drConfFile", "cd", "deep-retrieval", [
        ("data_path", str, REQ), ("model_path", str, REQ), ("mapping_path", str, REQ), ("thread_number", int, REQ),
        ("train_batch_size", int, REQ), ("eval_batch_size", int, REQ), ("num_layer", int, REQ), ("num_node", int, REQ),
        ("num_path_per_item", int, REQ), ("seq_len", int, REQ), ("min_seq_len", int, REQ), ("split_ratio", float, REQ),
        ("initialize_mapping", _bool, REQ), ("candidate_path_num", int, REQ), ("iteration_num", int, REQ),
        ("decay_factor", float, REQ), ("penalty_factor", float, REQ), ("penalty_poly_order", int, REQ), ("train_mode", str, REQ)]),
}
# JTM reuses the TDM tables for its first two stages (jtm/JTMInitializeTree.scala:21, JTMTrainDeepModel.scala:25)
TASKS["JTMInitializeTree"] = ("jtmConfFile", "init", "jtm", TASKS["TDMInitializeTree"][3])
TASKS["JTMTrainDeepModel"] = ("jtmConfFile", "model", "jtm", TASKS["TDMTrainDeepModel"][3])


def task_params(task, conf_path):
    """What `<Task> --<x>ConfFile conf_path` reads: the typed parameters of the task (getParameters of the reference).
    Adds `use_mask` (deep_model == "din") where the reference derives it and resolves thread_number (0 -> all cores)."""
    flag, prefix, _, table = TASKS[task]
    conf = read_conf(conf_path, prefix)
    out = {}
    for key, typ, default in table:
        if default is REQ:
            out[key] = typ(get_or_stop(conf, key))
        elif default is OPT:
            out[key] = typ(conf[key]) if key in conf else None
        else:
            out[key] = typ(conf.get(key, default))
    if "deep_model" in out:
        out["use_mask"] = out["deep_model"] == "din"
    if "thread_number" in out:
        out["thread_number"] = core_number(out["thread_number"])            # Property.configLocal -> Engine.setCoreNumber
    if "layer_negative_counts" in out:
        out["layer_negative_counts_list"] = [int(x) for x in out["layer_negative_counts"].split(",")]   # NegativeSampler.scala:27
    return out
