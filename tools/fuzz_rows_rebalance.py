#!/usr/bin/env python3
"""Randomised sweeps for the two device paths round 6 rewrote (GPU):
  rows      dm_din_forward through the general-rows split kernels (static-history instances L = 8 / 10 / 16 and the generic kernel): against the
            fp32-input MFMA kernel within the stated tolerance, run-to-run identical, prefix-stable (a row's score does not depend on its tile
            mates or its position in the tile);
  rebalance dm_jtm_rebalance_all / dm_otm_rebalance_all on the device (dev_sort.hip.inc: own radix sort and compaction) against the per-parent
            host logic (DM_JTM_REBALANCE=host): item for item, with crowded ties, NaN, signed zeros, capacities too small for everyone.
  python tools/fuzz_rows_rebalance.py [n_rows_configs=200] [n_rebalance_configs=200] [seed=1]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dismember_amd import Engine                      # noqa: E402
from dismember_amd import _native as N                # noqa: E402

n_rows, n_rb, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), (int(sys.argv[2]) if len(sys.argv) > 2 else 200), (int(sys.argv[3]) if len(sys.argv) > 3 else 1)
rng = np.random.default_rng(seed)
t0 = time.perf_counter()
bad = 0
engines = {}
for it in range(n_rows):
    E = int(rng.choice([32, 64, 128]))
    depth = int(rng.integers(6, 15)); ni = (1 << (depth + 1)) - 1
    key = (E, depth)
    if key not in engines:
        if len(engines) >= 6:
            engines.pop(next(iter(engines))).close()
        e = Engine(0); e.load_weights_din_synthetic(E, ni, int(rng.integers(1, 1 << 30)), tree_depth=depth, rho=float(rng.choice([0.0, 0.9])))
        engines[key] = e
    eng = engines[key]
    L = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 10, 11, 13, 16]))
    B = int(rng.choice([1, 5, 16, 17, 255, 4096, 40000]))
    codes = rng.integers(0, ni, B).astype(np.int32); codes[rng.random(B) < 0.02] = -1
    seqs = rng.integers(0, ni, (B, L)).astype(np.int32); seqs[rng.random((B, L)) < float(rng.choice([0.0, 0.2, 0.7]))] = -1
    pad = np.flatnonzero(seqs.reshape(-1) < 0).astype(np.int32) if rng.random() < 0.7 else None       # Mask.scala: pads masked, or not (use_mask false)
    eng.set_scorer_mode("auto"); a = eng.din_forward(codes, seqs, pad); b = eng.din_forward(codes, seqs, pad)
    eng.set_scorer_mode("f32"); r = eng.din_forward(codes, seqs, pad)
    eng.set_scorer_mode("auto")
    ok = np.array_equal(a, b) and np.isfinite(a).all() and (np.abs(a - r) <= 1e-5 + 1e-4 * np.abs(r)).all()
    n = int(rng.integers(1, B + 1)); o = int(rng.integers(0, min(B, 16)))
    pn = np.flatnonzero(seqs[:n].reshape(-1) < 0).astype(np.int32) if pad is not None else None
    po = np.flatnonzero(seqs[o:].reshape(-1) < 0).astype(np.int32) if pad is not None else None
    ok = ok and np.array_equal(eng.din_forward(codes[:n], seqs[:n], pn), a[:n]) and np.array_equal(eng.din_forward(codes[o:], seqs[o:], po), a[o:])
    if not ok:
        bad += 1
        print("ROWS MISMATCH", dict(E=E, depth=depth, L=L, B=B, masked=pad is not None), flush=True)
for e in engines.values():
    e.close()
print("rows: %d configurations, %d mismatches, %.0f s" % (n_rows, bad, time.perf_counter() - t0), flush=True)

t0 = time.perf_counter()
bad_rb = 0
eng = Engine(0)
for it in range(n_rb):
    f64 = rng.random() < 0.3
    gap = int(rng.choice([1, 2, 2, 2, 3, 4, 6]))
    old_level = int(rng.integers(0, 11))
    n = int(rng.choice([4096, 4097, 5000, 20000, 60000, 200000, 1000000]))             # (below 4096 items the library keeps the host logic)
    Cn, P = 1 << gap, 1 << old_level
    lo = P - 1
    item_node = (lo + rng.integers(0, P, n)).astype(np.int32)
    nv = int(rng.choice([2, 6, 1000]))                                    # few distinct weights = crowded ties
    w = (rng.integers(0, nv, (n, Cn)).astype(np.float32) - nv / 2) / 2.0
    if rng.random() < 0.5:
        w[:, 0] += 1.0                                                    # a popular child: cascading overflow
    if f64:
        w = w.astype(np.float64) + rng.integers(0, 3, (n, Cn)) * 2.0 ** -40
    if rng.random() < 0.3:
        w[rng.random((n, Cn)) < 0.05] = np.nan; w[rng.random((n, Cn)) < 0.05] = -0.0; w[rng.random((n, Cn)) < 0.05] = 0.0
    first = (item_node.astype(np.int64) << gap) + Cn - 1
    old_node = (first + rng.integers(0, Cn, n)).astype(np.int32)
    old_node[rng.random(n) < 0.1] = -7
    slack = float(rng.choice([0.7, 1.0, 1.02, 1.5]))
    max_assign = max(1, int(np.ceil(n / (P * Cn) * slack)))
    outs = {}
    for mode in ("device", "host"):
        os.environ["DM_JTM_REBALANCE"] = mode
        out = np.empty(n, np.int32)
        fn = N.lib().dm_otm_rebalance_all if f64 else N.lib().dm_jtm_rebalance_all
        eng._chk(fn(eng._h, w.ctypes.data_as(C.POINTER(C.c_double) if f64 else N.f32p), old_node.ctypes.data_as(N.i32p), item_node.ctypes.data_as(N.i32p),
                    n, old_level, old_level + gap, max_assign, out.ctypes.data_as(N.i32p)))
        outs[mode] = out
    del os.environ["DM_JTM_REBALANCE"]
    if not np.array_equal(outs["device"], outs["host"]):
        bad_rb += 1
        print("REBALANCE MISMATCH", dict(f64=f64, gap=gap, old_level=old_level, n=n, nv=nv, slack=slack, diff=int((outs["device"] != outs["host"]).sum())), flush=True)
eng.close()
print("rebalance: %d configurations, %d mismatches, %.0f s" % (n_rb, bad_rb, time.perf_counter() - t0), flush=True)
sys.exit(1 if bad or bad_rb else 0)
