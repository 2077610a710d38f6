"""Round 6: the general-rows scorer has a counted-load instance per common history length (dm_din_rows_split_l_kernel<E, L>, L = 8 / 10 / 16:
indices staged in LDS one tile ahead, inline-assembly gathers with explicit vmcnt waits) beside the generic kernel.  Both run the same
arithmetic in the same order; a register hazard in the counted loads shows up as run-to-run or prefix-dependent differences, which is
what these cases look for (tools/rows_determinism_probe.py is the long form)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from dismember_amd import Engine
E, L, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
depth = 13; ni = (1 << (depth + 1)) - 1
eng = Engine(0)
eng.load_weights_din_synthetic(E, ni, 7, tree_depth=depth, rho=0.9)
rng = np.random.default_rng(100 * E + L)
B = 30011
codes = rng.integers(0, ni, B).astype(np.int32); codes[::97] = -1
seqs = rng.integers(0, ni, (B, L)).astype(np.int32); seqs[rng.random((B, L)) < 0.2] = -1; seqs[5] = -1
np.save(out, eng.din_forward(codes, seqs))
eng.close()
"""


def _inputs(E, L):
    depth = 13; ni = (1 << (depth + 1)) - 1
    rng = np.random.default_rng(100 * E + L)
    B = 30011
    codes = rng.integers(0, ni, B).astype(np.int32); codes[::97] = -1
    seqs = rng.integers(0, ni, (B, L)).astype(np.int32); seqs[rng.random((B, L)) < 0.2] = -1; seqs[5] = -1
    return depth, ni, codes, seqs


@pytest.mark.parametrize("E,L", [(128, 10), (64, 8), (32, 16), (128, 16)])
def test_static_history_kernel_is_deterministic_prefix_stable_and_equals_the_generic_kernel(tmp_path, E, L):
    from dismember_amd import Engine
    depth, ni, codes, seqs = _inputs(E, L)
    eng = Engine(0)
    eng.load_weights_din_synthetic(E, ni, 7, tree_depth=depth, rho=0.9)
    full = eng.din_forward(codes, seqs)
    assert np.isfinite(full).all()
    assert np.array_equal(eng.din_forward(codes, seqs), full)                       # run to run
    for n in (1, 15, 16, 17, 4097):                                                 # a row's score does not depend on its tile mates
        assert np.array_equal(eng.din_forward(codes[:n], seqs[:n]), full[:n]), n
    assert np.array_equal(eng.din_forward(codes[5:], seqs[5:]), full[5:])           # nor on its position in the tile
    eng.set_scorer_mode("f32")
    ref = eng.din_forward(codes, seqs)
    eng.close()
    assert np.abs(full - ref).max() <= 1e-5 + 1e-4 * np.abs(ref).max()
    # the generic kernel (DM_ROWS_GENERIC=1 is read once per process): same arithmetic, same order -> the same bits
    out = str(tmp_path / "generic.npy")
    env = dict(os.environ, DM_ROWS_GENERIC="1")
    subprocess.run([sys.executable, "-c", _CHILD % ROOT, str(E), str(L), out], env=env, check=True, timeout=600)
    assert np.array_equal(np.load(out), full)


def test_bench_gpus_flag_refuses_a_job_larger_than_the_node():
    """`python bench.py --gpus N` with fewer than N visible devices exits non-zero with a message instead of printing a smaller job's line
    under that label (round-5 verdict, next #2); skipped on a node that really has that many devices."""
    from dismember_amd import _native
    import ctypes as C
    n = C.c_int(0)
    assert _native.lib().dm_device_count(C.byref(n)) == 0
    want = n.value + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DM_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""


_CHILD64 = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from dismember_amd import Engine
w, codes, seqs, pad = (np.load(sys.argv[i]) for i in (1, 2, 3, 4)); out = sys.argv[5]
E, L, NI = int(sys.argv[6]), seqs.shape[1], int(sys.argv[7])
eng = Engine(0); eng.load_weights_din(w, E, NI)
np.save(out, eng.din_forward(codes, seqs, pad))
eng.close()
"""


@pytest.mark.parametrize("E,L", [(16, 10), (64, 5), (128, 10), (32, 20)])
def test_f64_batch_forward_on_the_matrix_pipe_equals_the_scalar_kernel_and_the_oracle(tmp_path, E, L):
    """dm_din_forward on an f64 model: batches of >= 256 rows take the training kernel's forward half (v_mfma_f64_16x16x4_f64, fragments
    built on first use on a handle without training state); DM_FWD64_SCALAR=1 keeps the one-wave-per-row kernel.  Same function: both
    within 1e-10 / 1e-9 of the fp64 oracle, and of each other to summation order (1e-12 relative)."""
    from dismember_amd import Engine
    from oracle import pyoracle as po
    po.build()
    rng = np.random.default_rng(9000 + E + L)
    NI, B = 2047, 3001
    n = NI * E + 3 * E * E + 2 * E + 1
    w = (rng.standard_normal(n) * 0.2).astype(np.float64)
    codes = rng.integers(0, NI, B).astype(np.int32); codes[::53] = -1
    seqs = rng.integers(0, NI, (B, L)).astype(np.int32); seqs[rng.random((B, L)) < 0.25] = -1; seqs[7] = -1
    pad = np.flatnonzero(seqs.reshape(-1) < 0).astype(np.int32)
    ref = po.Din(w, E, L, NI).forward(codes, seqs, pad)
    eng = Engine(0); eng.load_weights_din(w, E, NI)
    fast = eng.din_forward(codes, seqs, pad)
    small = eng.din_forward(codes[:100], seqs[:100], np.flatnonzero(seqs[:100].reshape(-1) < 0).astype(np.int32))     # < 256 rows: the scalar kernel
    eng.close()
    assert (np.abs(fast - ref) <= 1e-10 + 1e-9 * np.abs(ref)).all()
    assert (np.abs(small - ref[:100]) <= 1e-10 + 1e-9 * np.abs(ref[:100])).all()
    files = [str(tmp_path / f) for f in ("w.npy", "c.npy", "s.npy", "p.npy", "o.npy")]
    for f, a in zip(files, (w, codes, seqs, pad)):
        np.save(f, a)
    env = dict(os.environ, DM_FWD64_SCALAR="1")
    subprocess.run([sys.executable, "-c", _CHILD64 % ROOT] + files + [str(E), str(NI)], env=env, check=True, timeout=600)
    scalar = np.load(files[4])
    assert (np.abs(fast - scalar) <= 1e-12 * (1.0 + np.abs(scalar))).all()
