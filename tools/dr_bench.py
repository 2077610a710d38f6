#!/usr/bin/env python3
"""Deep-Retrieval serving micro-benchmark (BASELINE config 5 shape): D=3, K=1000, beam=50, E=128, L=10.
Times dm_dr_beam_search_dev (and dm_dr_recommend_dev) with inputs resident in HBM; prints per-kernel event time."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--users", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--K", type=int, default=1000)
    ap.add_argument("--D", type=int, default=3)
    ap.add_argument("--beam", type=int, default=50)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--embed", type=int, default=128)
    ap.add_argument("--seq-len", type=int, default=10)
    ap.add_argument("--rerank", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="model and arithmetic type (f64 = the reference's)")
    a = ap.parse_args()
    E, L, K, D, U = a.embed, a.seq_len, a.K, a.D, a.users
    rng = np.random.default_rng(synth.SEED)
    eng = Engine(0)
    t0 = time.perf_counter()
    eng.dr_load_model_synthetic(E, L, K, D, a.items, synth.SEED, scale=0.05, rerank=bool(a.rerank), dtype=np.float64 if a.dtype == "f64" else np.float32)
    t_model = time.perf_counter() - t0
    seqs = rng.integers(0, a.items, size=(U, L)).astype(np.int32)
    seqs[rng.random((U, L)) < 0.15] = -1
    d_seq = eng.dev_alloc(U * L * 4); eng.h2d(d_seq, seqs)
    d_paths = eng.dev_alloc(U * a.beam * D * 4); d_probs = eng.dev_alloc(U * a.beam * 8); d_cnt = eng.dev_alloc(U * 4)
    out = {"model_load_s": t_model}
    import ctypes as C
    from dismember_amd import _native as N
    eng.dr_beam_search_dev(d_seq, U, a.beam, d_paths, d_probs, d_cnt)
    eng.synchronize(); eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.dr_beam_search_dev(d_seq, U, a.beam, d_paths, d_probs, d_cnt)
    eng.synchronize()
    dt = time.perf_counter() - t0
    nl, ms = eng.timing_get()
    out["beam_search"] = {"users_per_s": U * a.steps / dt, "ms_per_step": dt / a.steps * 1e3, "device_ms_per_step": ms / a.steps}
    # breakdown pass: an event pair around every launch (DM_DR_TIME_LAUNCHES=1, read per call) — the pairs drain the GPU between the kernels,
    # so this pass is slower than the timed one above and only its per-kernel times are kept
    os.environ["DM_DR_TIME_LAUNCHES"] = "1"
    eng.synchronize(); eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.dr_beam_search_dev(d_seq, U, a.beam, d_paths, d_probs, d_cnt)
    eng.synchronize()
    dtb = time.perf_counter() - t0
    os.environ.pop("DM_DR_TIME_LAUNCHES", None)
    nl, ms = eng.timing_get()
    out["beam_search"].update({"kernel_ms_per_step": ms / a.steps, "launches_per_step": nl / a.steps, "ms_per_step_with_per_launch_events": dtb / a.steps * 1e3})
    per = {}
    for kind, name in ((0, "gemm / single kernel"), (11, "layer0"), (12, "stats_d1"), (13, "select_d1"), (14, "stats_d2"), (15, "select_d2"), (23, "select_d1_todo_pass"), (25, "select_d2_todo_pass")):
        n_, ms_ = eng.timing_get_kind(kind)
        if n_:
            per[name] = round(ms_ / a.steps, 4)
    out["beam_search"]["kernel_ms_by_launch"] = per
    slow = C.c_ulonglong(0)
    N.lib().dm_debug_dr_slow_layers.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    N.lib().dm_debug_dr_slow_layers(eng._h, C.byref(slow), 1)
    out["beam_search"]["exact_path_layers_per_user"] = slow.value / float(U * (a.steps + 1))
    try:
        N.lib().dm_debug_dr_wave_fallbacks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
        two = (C.c_ulonglong * 2)()
        N.lib().dm_debug_dr_wave_fallbacks(eng._h, two, 1)
        out["beam_search"]["wave_cut_fallback_layers_per_user"] = two[0] / float(U * (a.steps + 1))
        out["beam_search"]["wave_cut_fallback_reasons_hex"] = hex(two[1])      # bytes: [1] degenerate sums, [2] score floor, [3] blocks, [4] too few, [5] too many
    except AttributeError:
        pass
    if a.rerank:
        t0 = time.perf_counter()
        pi = synth.dr_path_items_fast(synth.make_dr_paths(a.items, K, D, 2, rng), K)
        eng.dr_load_path_items(*pi)
        out["path_table_s"] = time.perf_counter() - t0
        out["paths"] = int(len(pi[0]))
        d_ids = eng.dev_alloc(U * a.topk * 4); d_sc = eng.dev_alloc(U * a.topk * 8)
        eng.dr_recommend_dev(d_seq, U, a.beam, a.topk, d_ids, d_sc, d_cnt)
        eng.synchronize(); eng.timing_reset()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.dr_recommend_dev(d_seq, U, a.beam, a.topk, d_ids, d_sc, d_cnt)
        eng.synchronize()
        dt = time.perf_counter() - t0
        nl, ms = eng.timing_get()
        cnt = np.empty(U, np.int32); eng.d2h(cnt, d_cnt)
        out["recommend"] = {"users_per_s": U * a.steps / dt, "ms_per_step": dt / a.steps * 1e3, "kernel_ms_per_step": ms / a.steps,
                            "mean_recs": float(cnt.mean())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
