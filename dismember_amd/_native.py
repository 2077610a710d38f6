"""ctypes binding of libdismember_hip.so (the C ABI of include/dismember_hip.h).

There is no Python or CPU implementation behind this module: if the shared
library is missing or no HIP device is usable, calls fail loudly.
"""
import ctypes as C
import os
import shutil
import time
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
# DM_LIB_PATH: a probe / development build of the same library (tools/build_probe.sh) instead of the in-tree product build
LIB_PATH = os.environ.get("DM_LIB_PATH") or os.path.join(_DIR, "libdismember_hip.so")
SRC_DIR = os.path.join(_DIR, "csrc")
INCLUDE_DIR = os.path.join(os.path.dirname(_DIR), "include")

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)


class AdamOpts(C.Structure):
    _fields_ = [("lr", C.c_double), ("lr_decay", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double)]


class DrModel(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("on_device", C.c_int32), ("embed", C.c_int32), ("seq_len", C.c_int32),
                ("num_node", C.c_int32), ("num_layer", C.c_int32), ("num_item", C.c_int64),
                ("layer_emb", C.c_void_p), ("layer_w", C.POINTER(C.c_void_p)), ("layer_b", C.POINTER(C.c_void_p)),
                ("rerank_emb", C.c_void_p), ("rerank_w", C.c_void_p), ("rerank_b", C.c_void_p),
                ("softmax_w", C.c_void_p), ("softmax_b", C.c_void_p)]


class SampleOpts(C.Structure):
    _fields_ = [("start_level", C.c_int), ("with_prob", C.c_int), ("tolerance", C.c_int), ("use_mask", C.c_int), ("seed", C.c_uint64)]


class OtmTrainOpts(C.Structure):
    _fields_ = [("beam", C.c_int), ("leaf_level", C.c_int), ("use_mask", C.c_int), ("target_mode", C.c_int)]


class SearchOpts(C.Structure):
    _fields_ = [("beam", C.c_int), ("topk", C.c_int), ("use_mask", C.c_int), ("widen_consumed", C.c_int)]


# name -> (restype, argtypes); also the list of symbols include/dismember_hip.h declares
SIGNATURES = {
    "dm_version": (C.c_int, []),
    "dm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dm_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "dm_destroy": (C.c_int, [C.c_void_p]),
    "dm_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "dm_last_error": (C.c_char_p, [C.c_void_p]),
    "dm_synchronize": (C.c_int, [C.c_void_p]),
    "dm_load_tree_tdm": (C.c_int, [C.c_void_p, i32p, i32p, u8p, C.c_int64, C.c_int]),
    "dm_load_tree_file": (C.c_int, [C.c_void_p, C.c_char_p]),
    "dm_load_id_maps": (C.c_int, [C.c_void_p, i32p, i32p, C.c_int64]),
    "dm_tdm_id_to_code": (C.c_int, [C.c_void_p, i32p, C.c_int, i32p, i32p, C.POINTER(C.c_int)]),
    "dm_level_start": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dm_load_weights_din": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64]),
    "dm_din_forward": (C.c_int, [C.c_void_p, i32p, i32p, i32p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "dm_tdm_beam_search": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.POINTER(SearchOpts), i64p, i32p, i32p,
                                     f32p, i32p]),
    "dm_tdm_beam_search_trace": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.POINTER(SearchOpts), i32p, f32p,
                                           i32p, C.c_int, i32p, f32p, i32p]),
    "dm_otm_beam_search": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, f32p, i32p]),
    "dm_otm_beam_search_trace": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, f32p, i32p, C.c_int,
                                           i32p, f32p, i32p]),
    "dm_tdm_bruteforce_topk": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, f32p, i32p]),
    "dm_jtm_child_weights": (C.c_int, [C.c_void_p, i64p, i32p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, f32p]),
    "dm_jtm_cache_rows": (C.c_int, [C.c_void_p, i64p, i32p, C.c_int64, C.c_int]),
    "dm_jtm_cache_rows_range": (C.c_int, [C.c_void_p, i64p, i32p, C.c_int64, C.c_int, C.c_int64, C.c_int64]),
    "dm_jtm_shard_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, i64p, i64p]),
    "dm_jtm_child_weights_cached": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
    "dm_jtm_last_step_seconds": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dm_jtm_optimize_cached": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.POINTER(C.c_double)]),
    "dm_jtm_optimize_all": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.POINTER(C.c_double)]),
    "dm_jtm_optimize_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "dm_jtm_step_cached": (C.c_int, [C.c_void_p, i32p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p]),
    "dm_jtm_rebalance": (C.c_int, [C.c_void_p, f32p, i32p, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_int, i32p]),
    "dm_jtm_rebalance_all": (C.c_int, [C.c_void_p, f32p, i32p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p]),
    "dm_otm_rebalance_all": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), i32p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p]),
    "dm_otm_child_weights": (C.c_int, [C.c_void_p, i64p, i32p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_double)]),
    "dm_otm_rebalance": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), i32p, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_int, i32p]),
    "dm_otm_train_batch": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, i64p, i32p, C.POINTER(OtmTrainOpts), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "dm_otm_pseudo_targets": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, i64p, i32p, C.POINTER(OtmTrainOpts), i32p, C.POINTER(C.c_double), i32p]),
    "dm_otm_train_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "dm_train_init": (C.c_int, [C.c_void_p, C.POINTER(AdamOpts)]),
    "dm_train_forward_backward": (C.c_int, [C.c_void_p, i32p, i32p, i32p, C.c_int64, f32p, C.c_int64, C.c_int, f32p]),
    "dm_adam_step": (C.c_int, [C.c_void_p, C.c_float]),
    "dm_adam_last_step_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dm_train_last_loss": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "dm_train_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "dm_train_dense_block": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), i64p]),
    "dm_train_export_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, i64p]),
    "dm_train_add_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "dm_tdm_set_node_probs": (C.c_int, [C.c_void_p, i32p, f32p, C.c_int64]),
    "dm_tdm_make_train_batch": (C.c_int, [C.c_void_p, i32p, i32p, C.c_int64, C.c_int, i32p, C.c_int, C.POINTER(SampleOpts),
                                          i32p, i32p, C.POINTER(C.c_uint32), f32p, C.c_int64, i64p]),
    "dm_tdm_sample_train_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, i32p, C.c_int,
                                                C.POINTER(SampleOpts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, i64p]),
    "dm_train_forward_backward_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                f32p]),
    "dm_train_forward_backward_grouped_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                        C.c_int, f32p]),
    "dm_dr_load_model": (C.c_int, [C.c_void_p, C.POINTER(DrModel)]),
    "dm_dr_load_path_items": (C.c_int, [C.c_void_p, i32p, C.c_int64, i64p, i32p]),
    "dm_dr_beam_search": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, i32p, C.POINTER(C.c_double), i32p]),
    "dm_dr_recommend": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, i32p, C.POINTER(C.c_double), i32p]),
    "dm_dr_beam_search_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dm_dr_recommend_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "dm_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dm_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dm_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dm_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dm_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dm_tdm_beam_search_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(SearchOpts),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dm_otm_beam_search_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dm_fill_normal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_uint64]),
    "dm_fill_normal_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_uint64]),
    "dm_fill_tree_normal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64]),
    "dm_load_weights_din_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64]),
    "dm_load_weights_din_dev_f64": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64]),
    "dm_save_model": (C.c_int, [C.c_void_p, C.c_char_p]),
    "dm_load_model": (C.c_int, [C.c_void_p, C.c_char_p]),
    "dm_set_scorer_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "dm_get_scorer_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dm_otm_beam_search_f64": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, C.POINTER(C.c_double), i32p]),
    "dm_otm_beam_search_trace_f64": (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, C.POINTER(C.c_double), i32p,
                                               C.c_int, i32p, C.POINTER(C.c_double), i32p]),
    "dm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dm_comm_create_rccl": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "dm_comm_create_tcp": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dm_comm_create_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "dm_comm_destroy": (C.c_int, [C.c_void_p]),
    "dm_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "dm_comm_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dm_comm_barrier": (C.c_int, [C.c_void_p]),
    "dm_comm_allreduce_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]),
    "dm_comm_all_gather_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "dm_comm_attach": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dm_comm_all_gather_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "dm_train_sync_gradients": (C.c_int, [C.c_void_p]),
    "dm_train_sync_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "dm_allreduce_grads": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dm_kernel_timing_reset": (C.c_int, [C.c_void_p]),
    "dm_kernel_timing_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "dm_last_beam_kernel": (C.c_char_p, [C.c_void_p]),
    "dm_kernel_timing_get_kind": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "dm_last_scored_rows": (C.c_int, [C.c_void_p, i64p]),
}


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(SRC_DIR, f) for f in sorted(os.listdir(SRC_DIR))]
    srcs.append(os.path.join(INCLUDE_DIR, "dismember_hip.h"))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-pthread", "-shared", "-fPIC", "-o", LIB_PATH,
           os.path.join(SRC_DIR, "dm_hip.hip"), "-I" + os.path.join(rocm, "include"), "-L" + os.path.join(rocm, "lib"),
           "-lrccl", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd))
    t_start = time.time()
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    cmd[cmd.index("-o") + 1] = tmp                      # (never a half-written library under the real name)
    try:
        subprocess.check_call(cmd)
        os.utime(tmp, (t_start, t_start))               # a source edited WHILE the compiler ran is newer than the library: next build() rebuilds
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdismember_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
