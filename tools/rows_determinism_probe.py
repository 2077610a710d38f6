import sys, numpy as np
sys.path.insert(0, '/root/repo')
from dismember_amd import Engine, synth
for E in (32, 64, 128):
    depth = 14; ni = (1 << (depth + 1)) - 1
    eng = Engine(0)
    eng.load_weights_din_synthetic(E, ni, 7, tree_depth=depth, rho=0.9)
    rng = np.random.default_rng(3)
    for L in (1, 3, 8, 10, 16):
        B = 40000
        codes = rng.integers(0, ni, B).astype(np.int32); codes[::97] = -1
        seqs = rng.integers(0, ni, (B, L)).astype(np.int32); seqs[rng.random((B, L)) < 0.2] = -1
        eng.set_scorer_mode("f32"); ref = eng.din_forward(codes, seqs)
        eng.set_scorer_mode("auto"); full = eng.din_forward(codes, seqs)
        err = np.abs(full - ref).max()
        bad = []
        for n in (1, 15, 16, 17, 1000, 33333):
            part = eng.din_forward(codes[:n], seqs[:n])
            if not np.array_equal(part, full[:n]): bad.append((n, int((part != full[:n]).sum())))
        off = eng.din_forward(codes[5:], seqs[5:])
        if not np.array_equal(off, full[5:]): bad.append(("off5", int((off != full[5:]).sum())))
        again = eng.din_forward(codes, seqs)
        print(E, L, "max|split-f32|", err, "rerun identical", np.array_equal(again, full), "prefix mismatches", bad, flush=True)
    eng.close()
