"""User-grouped training step (dm_train_forward_backward_grouped_dev, train_grouped_f64.hip.inc) — the OTM trainer's level batch:
MiniBatch.batchTransform (otm/src/main/scala/com/mass/otm/dataset/MiniBatch.scala:17-40) replicates a user's history over the
user's candidate nodes and LocalOptimizer.trainBatch (otm/.../optim/LocalOptimizer.scala:111-131) trains on the rows as DIN[Double].
The grouped kernels evaluate the per-user form of the same function; loss and every gradient are held to the fp64 oracle's
backward on the EXPANDED rows (1e-10 / 1e-9) and to the plain-rows kernel."""
import numpy as np
import pytest

from helpers import random_din_weights

pytestmark = pytest.mark.gpu


def _grouped_batch(rng, NI, U, n, L, pad_p=0.2):
    seq = rng.integers(0, NI, (U, L)).astype(np.int32)
    seq[rng.random((U, L)) < pad_p] = -1
    seq[0] = -1                                   # a user whose whole history is padding
    if U > 2:
        seq[2, :] = seq[2, 0] if seq[2, 0] >= 0 else 7     # one key repeated L times
    codes = rng.integers(0, NI, (U, n)).astype(np.int32)
    codes[rng.random((U, n)) < 0.02] = -1         # empty candidate slots (otm_batch_kernel: slot >= count)
    y = rng.random((U, n)).astype(np.float32)     # pseudo targets are clipped sums in [0, 1]
    y[rng.random((U, n)) < 0.6] = 0.0
    return seq, codes, y


def _expand(seq, codes, use_mask=True):
    U, L = seq.shape
    n = codes.shape[1]
    rseq = np.repeat(seq, n, axis=0)
    pad = np.flatnonzero(rseq.reshape(-1) == -1).astype(np.int32) if use_mask else np.zeros(0, np.int32)
    return codes.reshape(-1), rseq, pad


@pytest.mark.parametrize("E,NI,U,n,L", [(32, 255, 9, 40, 10), (128, 1023, 6, 400, 10), (16, 127, 5, 33, 16), (64, 511, 7, 17, 3),
                                        (128, 2047, 40, 50, 13)])
def test_grouped_step_vs_oracle_f64(oracle, E, NI, U, n, L):
    from dismember_amd import Engine
    rng = np.random.default_rng(E + U + n)
    w = random_din_weights(rng, E, NI, std=0.2, bias_std=0.2, dtype=np.float64)
    seq, codes, y = _grouped_batch(rng, NI, U, n, L)
    umask = np.zeros(U, np.uint32)
    for j in range(L):
        umask |= ((seq[:, j] == -1).astype(np.uint32) << np.uint32(j))
    eng = Engine(0)
    eng.load_weights_din(w, E, NI)
    eng.train_init(lr=1e-3)
    loss = eng.train_forward_backward_grouped(seq, umask, codes, y)
    g = eng.train_download("grad")
    rc, rseq, pad = _expand(seq, codes)
    odin = oracle.Din(w.copy(), E, L, NI)
    oloss, og = odin.train_grads(rc, rseq, pad, y.reshape(-1)) if (rc >= 0).all() else (None, None)
    # the plain-rows kernel on the expanded batch (handles code -1 rows the oracle's lookup would reject)
    eng2 = Engine(0)
    eng2.load_weights_din(w, E, NI)
    eng2.train_init(lr=1e-3)
    loss2 = eng2.train_forward_backward(rc, rseq, pad, y.reshape(-1))
    g2 = eng2.train_download("grad")
    assert abs(loss - loss2) <= 1e-10 + 1e-9 * abs(loss2), (loss, loss2)
    tol = 1e-10 * np.abs(g2).max() + 1e-9 * np.abs(g2)
    bad = np.abs(g - g2) > tol
    assert not bad.any(), (int(bad.sum()), np.flatnonzero(bad)[:10], float(np.abs(g - g2).max()))
    if og is not None:
        assert abs(loss - oloss) <= 1e-10 + 1e-9 * abs(oloss)
        assert (np.abs(g - og) <= 1e-10 * np.abs(og).max() + 1e-9 * np.abs(og)).all()
    # Adam on the grouped gradient: bit-exact restated update; untouched rows stay untouched
    touched = np.zeros(NI, bool); touched[rc[rc >= 0]] = True; touched[seq[seq >= 0]] = True
    assert (g[:NI * E].reshape(NI, E)[~touched] == 0).all()
    eng.adam_step(1.0)
    ref = w.copy()
    opt = oracle.Adam(ref.size, np.float64, lr=1e-3)
    opt.step(ref, g.copy())
    assert np.array_equal(eng.train_download("weights"), ref)
    # a second step on the moved weights (fragments refreshed), no mask this time: pads score 0 instead of -FLT_MAX
    loss_b = eng.train_forward_backward_grouped(seq, None, codes, y)
    eng2.adam_step(1.0)
    rc, rseq, pad0 = _expand(seq, codes, use_mask=False)
    loss2_b = eng2.train_forward_backward(rc, rseq, pad0, y.reshape(-1))
    gb, g2b = eng.train_download("grad"), eng2.train_download("grad")
    assert abs(loss_b - loss2_b) <= 1e-10 + 1e-9 * abs(loss2_b)
    assert (np.abs(gb - g2b) <= 1e-10 * np.abs(g2b).max() + 1e-9 * np.abs(g2b)).all(), float(np.abs(gb - g2b).max())
    eng.close(); eng2.close()


def test_grouped_step_oracle_direct(oracle):
    """No -1 candidate rows: the oracle's own backward on the expanded rows is the reference."""
    from dismember_amd import Engine
    E, NI, U, n, L = 128, 1023, 5, 64, 10
    rng = np.random.default_rng(3)
    w = random_din_weights(rng, E, NI, std=0.2, bias_std=0.2, dtype=np.float64)
    seq = rng.integers(0, NI, (U, L)).astype(np.int32)
    seq[1, :4] = -1
    codes = rng.integers(0, NI, (U, n)).astype(np.int32)
    y = (rng.random((U, n)) < 0.3).astype(np.float32)
    umask = np.zeros(U, np.uint32)
    for j in range(L):
        umask |= ((seq[:, j] == -1).astype(np.uint32) << np.uint32(j))
    eng = Engine(0)
    eng.load_weights_din(w, E, NI)
    eng.train_init(lr=1e-3)
    loss = eng.train_forward_backward_grouped(seq, umask, codes, y)
    g = eng.train_download("grad")
    rc, rseq, pad = _expand(seq, codes)
    oloss, og = oracle.Din(w.copy(), E, L, NI).train_grads(rc, rseq, pad, y.reshape(-1))
    assert abs(loss - oloss) <= 1e-10 + 1e-9 * abs(oloss), (loss, oloss)
    err = np.abs(g - og)
    assert (err <= 1e-10 * np.abs(og).max() + 1e-9 * np.abs(og)).all(), float(err.max())
    eng.close()


def test_grouped_step_f32_model_expands_to_rows(oracle):
    """An f32 model takes the plain-rows kernel behind the same entry point."""
    from dismember_amd import Engine
    E, NI, U, n, L = 32, 255, 6, 20, 10
    rng = np.random.default_rng(4)
    w = random_din_weights(rng, E, NI, std=0.2, bias_std=0.2)
    seq, codes, y = _grouped_batch(rng, NI, U, n, L)
    codes = np.abs(codes)
    umask = np.zeros(U, np.uint32)
    for j in range(L):
        umask |= ((seq[:, j] == -1).astype(np.uint32) << np.uint32(j))
    eng = Engine(0)
    eng.load_weights_din(w, E, NI)
    eng.train_init(lr=1e-3)
    loss = eng.train_forward_backward_grouped(seq, umask, codes, y)
    g = eng.train_download("grad")
    rc, rseq, pad = _expand(seq, codes)
    oloss, og = oracle.Din(w.copy(), E, L, NI).train_grads(rc, rseq, pad, y.reshape(-1))
    assert abs(loss - oloss) <= 1e-5 + 1e-4 * abs(oloss)
    assert (np.abs(g - og) <= 2e-5 * np.abs(og).max() + 1e-4 * np.abs(og)).all()
    eng.close()
