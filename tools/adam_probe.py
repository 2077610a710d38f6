import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
from helpers import random_din_weights
from dismember_amd import Engine
from oracle import pyoracle as po
rng = np.random.default_rng(1)
E, NI, B = 16, 8191, 300
w = np.load("tests/golden/din_f32.npy")
eng = Engine(0); eng.load_weights_din(w, E, NI); eng.train_init(lr=1e-3)
codes = rng.integers(0, NI, B).astype(np.int32); seqs = rng.integers(0, NI, (B, 10)).astype(np.int32)
y = (rng.random(B) < 0.3).astype(np.float32)
eng.train_forward_backward(codes, seqs, None, y)
g = eng.train_download("grad"); eng.adam_step(1.0)
w1 = eng.train_download("weights"); s1 = eng.train_download("s"); r1 = eng.train_download("r")
ref = w.copy(); opt = po.Adam(ref.size, np.float32, lr=1e-3); opt.step(ref, g.copy())
for name, a, b in (("w", w1, ref), ("s", s1, opt.s), ("r", r1, opt.r)):
    d = np.flatnonzero(a != b)
    print(name, "differs at", d.size, "of", a.size)
    for i in d[:5]:
        print("   i", i, "g", g[i], "gpu", repr(a[i]), "ref", repr(b[i]), "w0", w[i])
