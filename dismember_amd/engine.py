"""Thin object wrapper over the C ABI: one Engine == one dm_handle_t (device + stream)."""
import ctypes as C
import os

import numpy as np

from . import _native as N


class DismemberError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dismember_hip error %d: %s" % (code, msg))
        self.code = code


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


class Engine:
    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        rc = N.lib().dm_create(device_id, C.byref(self._h))
        if rc != 0:
            raise DismemberError(rc, (N.lib().dm_last_error(None) or b"").decode())
        self.E = None
        self.dtype = None

    def clone(self):
        """dm_clone: a second engine on the same device that reads this engine's tree, weights and derived copies (no copy is made)
        through its own stream and request buffers — the reference's cloneModule() workers (LocalOptimizer.scala:28-44).  Loading and
        training go through the owner; close the clones first."""
        c = Engine.__new__(Engine)
        c._h = C.c_void_p()
        self._chk(N.lib().dm_clone(self._h, C.byref(c._h)))
        c.E, c.dtype = self.E, self.dtype
        c._owner = self                      # keeps the owner alive for as long as the clone exists
        return c

    def close(self):
        if getattr(self, "_h", None):
            rc = N.lib().dm_destroy(self._h)
            if rc != 0:
                raise DismemberError(rc, (N.lib().dm_last_error(self._h) or b"").decode())
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise DismemberError(rc, (N.lib().dm_last_error(self._h) or b"").decode())

    # ---- loading
    def load_tree(self, codes, node_ids, is_leaf, max_level):
        codes, node_ids = _i32(codes), _i32(node_ids)
        is_leaf = np.ascontiguousarray(is_leaf, dtype=np.uint8)
        self.max_level = int(max_level)
        self._chk(N.lib().dm_load_tree_tdm(self._h, _p(codes, N.i32p), _p(node_ids, N.i32p), _p(is_leaf, N.u8p),
                                           codes.size, int(max_level)))

    def load_tree_file(self, path):
        """TDM.loadTree / TDMOp.initTree(treePbPath) (tdm/.../model/TDM.scala:50-52; DistTree.loadData, tdm/.../tree/DistTree.scala:40-87):
        a reference tree file -> device index + id maps + node probabilities, parsed inside the library (dm_load_tree_file)."""
        self._chk(N.lib().dm_load_tree_file(self._h, os.fsencode(path)))
        with open(path, "rb") as f:                    # max_level for the Python-side bookkeeping: the tree_meta record closes the file
            data = f.read()
        from . import tree_io
        self.max_level = tree_io.read_max_level(data)

    def load_id_maps(self, leaf_item_ids, leaf_codes):
        a, b = _i32(leaf_item_ids), _i32(leaf_codes)
        self._chk(N.lib().dm_load_id_maps(self._h, _p(a, N.i32p), _p(b, N.i32p), a.size))

    def load_weights_din(self, compact, E, num_index):
        w = np.ascontiguousarray(compact)
        if w.dtype == np.float32:
            dt = 0
        elif w.dtype == np.float64:
            dt = 1
        else:
            raise TypeError("weights must be float32 or float64")
        self._chk(N.lib().dm_load_weights_din(self._h, dt, int(E), int(num_index), w.ctypes.data_as(C.c_void_p), w.size))
        self.E, self.dtype, self.num_index = int(E), w.dtype, int(num_index)

    def load_weights_din_synthetic(self, E, num_index, seed, small=None, tree_depth=None, rho=0.0, std=0.05):
        """Build the compact DIN vector on the device (N(0, 0.05) table; `small` = host array holding
        [att.W ; l1.W ; l1.b ; l2.W ; l2.b], defaults to the reference init) and load it without a host copy."""
        n = num_index * E + 3 * E * E + 2 * E + 1
        d = self.dev_alloc(n * 4)
        if tree_depth is not None and rho > 0.0:
            assert num_index == (1 << (tree_depth + 1)) - 1
            self._chk(N.lib().dm_fill_tree_normal(self._h, d, int(E), int(tree_depth), float(rho), float(std), int(seed)))
        else:
            self._chk(N.lib().dm_fill_normal(self._h, d, num_index * E, 0.0, float(std), int(seed)))
        if small is None:
            rng = np.random.default_rng(int(seed))
            small = np.zeros(3 * E * E + 2 * E + 1, np.float32)
            small[:3 * E * E] = rng.standard_normal(3 * E * E, dtype=np.float32) * 0.05
            small[3 * E * E + E:3 * E * E + 2 * E] = rng.standard_normal(E, dtype=np.float32) * 0.05
        small = np.ascontiguousarray(small, dtype=np.float32)
        assert small.size == 3 * E * E + 2 * E + 1
        self._chk(N.lib().dm_memcpy_h2d(self._h, C.c_void_p(d.value + num_index * E * 4),
                                        small.ctypes.data_as(C.c_void_p), small.nbytes))
        self._chk(N.lib().dm_load_weights_din_dev(self._h, int(E), int(num_index), d, n))
        self.E, self.dtype, self.num_index = int(E), np.dtype(np.float32), int(num_index)
        self._synth_ptr, self._synth_n = d, n
        return d

    def load_weights_din_synthetic_f64(self, E, num_index, seed, small=None):
        """The fp64 counterpart (the reference's OTM model type): the table holds the f32 draws widened to double
        (dm_fill_normal_f64), the small matrices come from `small` (defaults to the reference init) as doubles."""
        n = num_index * E + 3 * E * E + 2 * E + 1
        d = self.dev_alloc(n * 8)
        self._chk(N.lib().dm_fill_normal_f64(self._h, d, num_index * E, 0.0, 0.05, int(seed)))
        if small is None:
            rng = np.random.default_rng(int(seed))
            small = np.zeros(3 * E * E + 2 * E + 1, np.float32)
            small[:3 * E * E] = rng.standard_normal(3 * E * E, dtype=np.float32) * 0.05
            small[3 * E * E + E:3 * E * E + 2 * E] = rng.standard_normal(E, dtype=np.float32) * 0.05
        small = np.ascontiguousarray(small, dtype=np.float64)
        assert small.size == 3 * E * E + 2 * E + 1
        self._chk(N.lib().dm_memcpy_h2d(self._h, C.c_void_p(d.value + num_index * E * 8),
                                        small.ctypes.data_as(C.c_void_p), small.nbytes))
        self._chk(N.lib().dm_load_weights_din_dev_f64(self._h, int(E), int(num_index), d, n))
        self.E, self.dtype, self.num_index = int(E), np.dtype(np.float64), int(num_index)
        self._synth_ptr, self._synth_n = d, n
        return d

    def save_model(self, path):
        """TDM.saveModel (T/model/TDM.scala:32-41): weights + index in one flat file (dm_save_model)."""
        self._chk(N.lib().dm_save_model(self._h, os.fsencode(path)))

    def load_model(self, path):
        """TDM.loadModel: replaces the handle's tree, id maps and weights with the checkpoint's."""
        self._chk(N.lib().dm_load_model(self._h, os.fsencode(path)))
        import struct
        with open(path, "rb") as f:
            hd = f.read(8 + 8 * 4 + 4 * 8)
        _, dtype, E, max_level, has_tree, _, _, _ = struct.unpack("<8i", hd[8:40])
        num_index, = struct.unpack("<q", hd[40:48])
        self.E, self.dtype, self.num_index = int(E), np.dtype(np.float64 if dtype == 1 else np.float32), int(num_index)
        if has_tree:
            self.max_level = int(max_level)

    def download_weights(self):
        """Host copy of the compact vector built by load_weights_din_synthetic (for the CPU oracle)."""
        out = np.empty(self._synth_n, self.dtype)
        self.d2h(out, self._synth_ptr)
        return out

    # ---- operator level
    def id_to_code(self, item_ids):
        ids = _i32(item_ids)
        codes = np.empty_like(ids)
        mask = np.empty_like(ids)
        nm = C.c_int(0)
        self._chk(N.lib().dm_tdm_id_to_code(self._h, _p(ids, N.i32p), ids.size, _p(codes, N.i32p), _p(mask, N.i32p),
                                            C.byref(nm)))
        return codes, mask[:nm.value].copy()

    def din_forward(self, codes, seqs, pad_flat_idx=None, L=None):
        codes = _i32(codes).ravel()
        seqs = _i32(seqs)
        B = codes.size
        if L is None:
            L = seqs.shape[-1] if seqs.ndim == 2 else seqs.size // max(B, 1)
        seqs = seqs.ravel()
        pad = _i32([] if pad_flat_idx is None else pad_flat_idx).ravel()
        out = np.empty(B, dtype=self.dtype)
        self._chk(N.lib().dm_din_forward(self._h, _p(codes, N.i32p), _p(seqs, N.i32p), _p(pad, N.i32p), pad.size, B, L,
                                         out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- beam search
    @staticmethod
    def _csr(consumed, U):
        if consumed is None:
            return None, None
        off = np.zeros(U + 1, np.int64)
        for u in range(U):
            off[u + 1] = off[u] + len(consumed[u])
        ids = _i32(np.concatenate([_i32(c) for c in consumed]) if off[U] > 0 else np.zeros(1, np.int32))
        return off, ids

    def tdm_beam_search(self, seq_item_ids, beam, topk, use_mask=True, consumed=None, widen_consumed=False, out=None):
        """out: optional (ids [U, topk] int32, scores [U, topk] float32, counts [U] int32) to fill — a serving loop reuses its result
        buffers (fresh 200 MB arrays page-fault on every call, inside the download)."""
        seq = _i32(seq_item_ids)
        if seq.ndim == 1:
            seq = seq[None, :]
        U, L = seq.shape
        opts = N.SearchOpts(int(beam), int(topk), int(bool(use_mask)), int(bool(widen_consumed)))
        off, cids = self._csr(consumed, U)
        if out is not None:
            ids, sc, cnt = out
            assert ids.shape == (U, topk) and ids.dtype == np.int32 and sc.shape == (U, topk) and sc.dtype == np.float32
            assert cnt.shape == (U,) and cnt.dtype == np.int32 and ids.flags.c_contiguous and sc.flags.c_contiguous
        else:
            ids = np.empty((U, topk), np.int32)
            sc = np.empty((U, topk), np.float32)
            cnt = np.empty(U, np.int32)
        self._chk(N.lib().dm_tdm_beam_search(self._h, _p(seq, N.i32p), U, L, C.byref(opts),
                                             None if off is None else _p(off, N.i64p),
                                             None if cids is None else _p(cids, N.i32p), _p(ids, N.i32p),
                                             _p(sc, N.f32p), _p(cnt, N.i32p)))
        return ids, sc, cnt

    def tdm_beam_search_trace(self, seq_item_ids, beam, topk, use_mask=True, max_levels=None):
        seq = _i32(seq_item_ids)
        if seq.ndim == 1:
            seq = seq[None, :]
        U, L = seq.shape
        if max_levels is None:
            max_levels = self.max_level + 2
        cap = max(32, ((2 * beam + 15) // 16) * 16)
        opts = N.SearchOpts(int(beam), int(topk), int(bool(use_mask)), 0)
        ids = np.empty((U, topk), np.int32)
        sc = np.empty((U, topk), np.float32)
        cnt = np.empty(U, np.int32)
        tc = np.zeros((U, max_levels, cap), np.int32)
        ts = np.zeros((U, max_levels, cap), np.float32)
        tn = np.zeros((U, max_levels), np.int32)
        self._chk(N.lib().dm_tdm_beam_search_trace(self._h, _p(seq, N.i32p), U, L, C.byref(opts), _p(ids, N.i32p),
                                                   _p(sc, N.f32p), _p(cnt, N.i32p), max_levels, _p(tc, N.i32p),
                                                   _p(ts, N.f32p), _p(tn, N.i32p)))
        return ids, sc, cnt, tc, ts, tn

    def otm_beam_search(self, seq_codes, beam, leaf_level, out=None):
        seq = _i32(seq_codes)
        if seq.ndim == 1:
            seq = seq[None, :]
        U, L = seq.shape
        if out is not None:
            ids, sc, cnt = out
            assert ids.shape == (U, 2 * beam) and ids.dtype == np.int32 and sc.shape == (U, 2 * beam) and sc.dtype == np.float32
            assert cnt.shape == (U,) and cnt.dtype == np.int32 and ids.flags.c_contiguous and sc.flags.c_contiguous
        else:
            ids = np.empty((U, 2 * beam), np.int32)
            sc = np.empty((U, 2 * beam), np.float32)
            cnt = np.empty(U, np.int32)
        self._chk(N.lib().dm_otm_beam_search(self._h, _p(seq, N.i32p), U, L, int(beam), int(leaf_level), _p(ids, N.i32p),
                                             _p(sc, N.f32p), _p(cnt, N.i32p)))
        return ids, sc, cnt

    def otm_beam_search_f64(self, seq_codes, beam, leaf_level, trace_levels=None):
        """CandidateSearcher.beamSearch in the reference's own arithmetic (DIN[Double]); f64 weights required.  With
        trace_levels: also every level's candidates (OTMTree.beamSearchNodes): (ids, scores, counts, tc, ts, tn)."""
        seq = _i32(seq_codes)
        if seq.ndim == 1:
            seq = seq[None, :]
        U, L = seq.shape
        f64p = C.POINTER(C.c_double)
        ids = np.empty((U, 2 * beam), np.int32)
        sc = np.empty((U, 2 * beam), np.float64)
        cnt = np.empty(U, np.int32)
        if trace_levels is None:
            self._chk(N.lib().dm_otm_beam_search_f64(self._h, _p(seq, N.i32p), U, L, int(beam), int(leaf_level), _p(ids, N.i32p),
                                                     sc.ctypes.data_as(f64p), _p(cnt, N.i32p)))
            return ids, sc, cnt
        cap = max(32, ((2 * beam + 15) // 16) * 16)
        tc = np.zeros((U, trace_levels, cap), np.int32)
        ts = np.zeros((U, trace_levels, cap), np.float64)
        tn = np.zeros((U, trace_levels), np.int32)
        self._chk(N.lib().dm_otm_beam_search_trace_f64(self._h, _p(seq, N.i32p), U, L, int(beam), int(leaf_level), _p(ids, N.i32p),
                                                       sc.ctypes.data_as(f64p), _p(cnt, N.i32p), int(trace_levels),
                                                       _p(tc, N.i32p), ts.ctypes.data_as(f64p), _p(tn, N.i32p)))
        return ids, sc, cnt, tc, ts, tn

    def otm_beam_search_trace(self, seq_codes, beam, leaf_level, trace_levels):
        """dm_otm_beam_search_trace (float scores; arithmetic per the scorer mode): (ids, scores, counts, tc, ts, tn)."""
        seq = _i32(seq_codes)
        if seq.ndim == 1:
            seq = seq[None, :]
        U, L = seq.shape
        cap = max(32, ((2 * beam + 15) // 16) * 16)
        ids = np.empty((U, 2 * beam), np.int32); sc = np.empty((U, 2 * beam), np.float32); cnt = np.empty(U, np.int32)
        tc = np.zeros((U, trace_levels, cap), np.int32); ts = np.zeros((U, trace_levels, cap), np.float32)
        tn = np.zeros((U, trace_levels), np.int32)
        self._chk(N.lib().dm_otm_beam_search_trace(self._h, _p(seq, N.i32p), U, L, int(beam), int(leaf_level), _p(ids, N.i32p),
                                                   _p(sc, N.f32p), _p(cnt, N.i32p), int(trace_levels), _p(tc, N.i32p),
                                                   _p(ts, N.f32p), _p(tn, N.i32p)))
        return ids, sc, cnt, tc, ts, tn

    def tdm_bruteforce_topk(self, seq_item_ids, topk, use_mask=True):
        seq = _i32(seq_item_ids)
        if seq.ndim == 1:
            seq = seq[None, :]
        U, L = seq.shape
        ids = np.empty((U, topk), np.int32)
        sc = np.empty((U, topk), np.float32)
        cnt = np.empty(U, np.int32)
        self._chk(N.lib().dm_tdm_bruteforce_topk(self._h, _p(seq, N.i32p), U, L, int(topk), int(bool(use_mask)),
                                                 _p(ids, N.i32p), _p(sc, N.f32p), _p(cnt, N.i32p)))
        return ids, sc, cnt

    # ---- training (row A12)
    def train_init(self, lr=1e-3, lr_decay=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
        o = N.AdamOpts(lr, lr_decay, beta1, beta2, eps)
        self._chk(N.lib().dm_train_init(self._h, C.byref(o)))

    def train_forward_backward(self, codes, seqs, pad_flat_idx, labels):
        codes = _i32(codes).ravel()
        seqs = _i32(seqs)
        B = codes.size
        L = seqs.shape[-1] if seqs.ndim == 2 else seqs.size // B
        seqs = seqs.ravel()
        pad = _i32([] if pad_flat_idx is None else pad_flat_idx).ravel()
        lab = np.ascontiguousarray(labels, np.float32).ravel()
        loss = C.c_float(0)
        self._chk(N.lib().dm_train_forward_backward(self._h, _p(codes, N.i32p), _p(seqs, N.i32p), _p(pad, N.i32p), pad.size,
                                                    _p(lab, N.f32p), B, L, C.byref(loss)))
        return self.train_last_loss() if self.dtype == np.float64 else loss.value

    def train_last_loss(self):
        """The loss of the last forward/backward in the model's precision (dm_train_last_loss)."""
        v = C.c_double(0)
        self._chk(N.lib().dm_train_last_loss(self._h, C.byref(v)))
        return v.value

    def train_sync_stats(self):
        o = (C.c_uint64 * 8)()
        self._chk(N.lib().dm_train_sync_stats(self._h, o))
        return {"nranks": int(o[0]), "transport": "rccl" if o[1] == 1 else "host", "rows_mine": int(o[2]), "rows_total": int(o[3]),
                "bytes_sent": int(o[4]), "bytes_recv": int(o[5]), "host_syncs": int(o[6])}

    def set_node_probs(self, codes, probs):
        """Node.probality per tree node (needed by sample_with_probability)."""
        c = _i32(codes); pr = np.ascontiguousarray(probs, np.float32)
        self._chk(N.lib().dm_tdm_set_node_probs(self._h, _p(c, N.i32p), _p(pr, N.f32p), c.size))

    def make_train_batch(self, seq_item_ids, target_item_ids, neg_counts, start_level=1, seed=0, use_mask=True, with_prob=False,
                         tolerance=20):
        """NegativeSampler.sample + MiniBatch.convert (sampled on the device): (codes [R], seqs [R, L], rowmask [R], labels [R])."""
        seq = _i32(seq_item_ids)
        tgt = _i32(target_item_ids).ravel()
        T, L = seq.shape
        neg = _i32(neg_counts)
        o = N.SampleOpts(int(start_level), int(bool(with_prob)), int(tolerance), int(bool(use_mask)), int(seed))
        n = C.c_int64(0)
        self._chk(N.lib().dm_tdm_make_train_batch(self._h, _p(seq, N.i32p), _p(tgt, N.i32p), T, L, _p(neg, N.i32p), neg.size,
                                                  C.byref(o), None, None, None, None, 0, C.byref(n)))
        R = n.value
        codes = np.empty(max(R, 1), np.int32); seqs = np.empty((max(R, 1), L), np.int32)
        mask = np.empty(max(R, 1), np.uint32); lab = np.empty(max(R, 1), np.float32)
        self._chk(N.lib().dm_tdm_make_train_batch(self._h, _p(seq, N.i32p), _p(tgt, N.i32p), T, L, _p(neg, N.i32p), neg.size,
                                                  C.byref(o), _p(codes, N.i32p), _p(seqs, N.i32p),
                                                  mask.ctypes.data_as(C.POINTER(C.c_uint32)), _p(lab, N.f32p), R, C.byref(n)))
        R = n.value
        return codes[:R], seqs[:R], mask[:R], lab[:R]

    def train_step_sampled(self, seq_item_ids, target_item_ids, neg_counts, start_level=1, seed=0, use_mask=True, with_prob=False,
                           tolerance=20):
        """convertBatch + trainBatch with the rows resident in HBM: targets and histories are uploaded (40 B per target),
        dm_tdm_sample_train_batch_dev expands them on the device straight into the buffers of
        dm_train_forward_backward_dev.  Returns the mean BCE loss of the expanded rows."""
        seq = _i32(seq_item_ids)
        tgt = _i32(target_item_ids).ravel()
        T, L = seq.shape
        neg = _i32(neg_counts)
        o = N.SampleOpts(int(start_level), int(bool(with_prob)), int(tolerance), int(bool(use_mask)), int(seed))
        n = C.c_int64(0)
        self._chk(N.lib().dm_tdm_sample_train_batch_dev(self._h, None, None, T, L, _p(neg, N.i32p), neg.size, C.byref(o), None, None,
                                                        None, None, 0, C.byref(n)))
        R = max(n.value, 1)
        al = lambda v: (v + 255) & ~255
        need = al(T * L * 4) + al(T * 4) + 2 * al(R * 4) + al(R * L * 4) + al(R * 4)     # the six sub-buffers as carved below
        if getattr(self, "_samp_bytes", 0) < need:
            if getattr(self, "_samp_buf", None):
                self.dev_free(self._samp_buf)
            self._samp_buf, self._samp_bytes = self.dev_alloc(need + need // 2), need + need // 2
        base = self._samp_buf.value
        d_seq = C.c_void_p(base); base += al(T * L * 4)
        d_tgt = C.c_void_p(base); base += al(T * 4)
        d_codes = C.c_void_p(base); base += al(R * 4)
        d_seqs = C.c_void_p(base); base += al(R * L * 4)
        d_mask = C.c_void_p(base); base += al(R * 4)
        d_lab = C.c_void_p(base)
        self.h2d(d_seq, seq); self.h2d(d_tgt, tgt)
        self._chk(N.lib().dm_tdm_sample_train_batch_dev(self._h, d_seq, d_tgt, T, L, _p(neg, N.i32p), neg.size, C.byref(o), d_codes,
                                                        d_seqs, d_mask, d_lab, R, C.byref(n)))
        if n.value == 0:
            return 0.0
        loss = C.c_float(0)
        self._chk(N.lib().dm_train_forward_backward_dev(self._h, d_codes, d_seqs, d_mask, d_lab, n.value, L, C.byref(loss)))
        return loss.value

    def train_forward_backward_grouped(self, seq_codes, user_mask, codes, labels):
        """One forward/backward over a user-grouped batch (dm_train_forward_backward_grouped_dev): seq_codes [U][L], user_mask [U]
        (bit j: position j masked) or None, codes / labels [U][n] — every user's n candidate rows share the user's history, as
        MiniBatch.batchTransform / transformWithMask build them.  Returns the mean BCE loss (model precision)."""
        seq = _i32(seq_codes)
        U, L = seq.shape
        codes = _i32(codes).reshape(U, -1)
        n = codes.shape[1]
        lab = np.ascontiguousarray(labels, np.float32).reshape(U, n)
        al = lambda v: (v + 255) & ~255
        need = al(U * L * 4) + al(U * 4) + 2 * al(U * n * 4)
        buf = self.dev_alloc(need)
        try:
            base = buf.value
            d_seq = C.c_void_p(base); base += al(U * L * 4)
            d_um = C.c_void_p(base); base += al(U * 4)
            d_codes = C.c_void_p(base); base += al(U * n * 4)
            d_lab = C.c_void_p(base)
            self.h2d(d_seq, seq); self.h2d(d_codes, codes); self.h2d(d_lab, lab)
            if user_mask is not None:
                self.h2d(d_um, np.ascontiguousarray(user_mask, np.uint32).ravel())
            loss = C.c_float(0)
            self._chk(N.lib().dm_train_forward_backward_grouped_dev(self._h, d_seq, d_um if user_mask is not None else None, d_codes, d_lab,
                                                                    U, n, L, C.byref(loss)))
            self.synchronize()
        finally:
            self.dev_free(buf)
        return self.train_last_loss() if self.dtype == np.float64 else loss.value

    @staticmethod
    def rowmask_to_flat(mask, L):
        """Row bit masks -> the flat index list Module.forward takes (Mask.scala:27-32)."""
        i, j = np.nonzero((mask[:, None] >> np.arange(L, dtype=np.uint32)[None, :]) & 1)
        return (i * L + j).astype(np.int32)

    # ---- multi-GPU exchange (comm.py)
    def attach_comm(self, comm):
        """comm: dismember_amd.comm.Comm, a raw dm_comm_t (comm.make_clique) or None."""
        self._comm = comm
        self._chk(N.lib().dm_comm_attach(self._h, None if comm is None else getattr(comm, "_c", comm)))

    def train_sync_gradients(self):
        """LocalOptimizer.syncGradients over the attached communicator (collective: every rank calls it)."""
        self._chk(N.lib().dm_train_sync_gradients(self._h))

    def comm_all_gather_dev(self, arr):
        """Var-size all-gather of DEVICE buffers over the attached communicator (dm_comm_all_gather_dev): uploads `arr`,
        gathers on the handle's stream (RCCL broadcasts, or host staging on the host transport), returns the concatenation."""
        a = np.ascontiguousarray(arr)
        if hasattr(self._comm, "world"):
            world = self._comm.world
        else:                                          # a raw dm_comm_t (comm.make_clique): ask the library
            r_, w_, t_ = C.c_int(0), C.c_int(1), C.c_int(0)
            self._chk(N.lib().dm_comm_rank(self._comm, C.byref(r_), C.byref(w_), C.byref(t_)))
            world = w_.value
        sizes = (C.c_uint64 * world)()
        d_send = self.dev_alloc(max(a.nbytes, 16))
        self.h2d(d_send, a)
        self._chk(N.lib().dm_comm_all_gather_dev(self._h, d_send, a.nbytes, None, 0, sizes))
        total = int(sum(sizes))
        d_recv = self.dev_alloc(max(total, 16))
        self._chk(N.lib().dm_comm_all_gather_dev(self._h, d_send, a.nbytes, d_recv, total, sizes))
        out = np.empty(total // a.itemsize, a.dtype)
        self.d2h(out, d_recv)
        self.dev_free(d_send); self.dev_free(d_recv)
        return out.reshape((-1,) + a.shape[1:])

    def adam_step(self, grad_scale=1.0):
        self._chk(N.lib().dm_adam_step(self._h, float(grad_scale)))

    def adam_last_step_rows(self):
        """(rows visited by the last Adam step, True when it took the active-rows path)."""
        r, a = C.c_uint64(0), C.c_int(0)
        self._chk(N.lib().dm_adam_last_step_rows(self._h, C.byref(r), C.byref(a)))
        return int(r.value), bool(a.value)

    def train_download(self, what="weights"):
        n = self.num_index * self.E + 3 * self.E * self.E + 2 * self.E + 1
        out = np.empty(n, self.dtype)          # the loaded dtype: fp64 models train in fp64
        self._chk(N.lib().dm_train_download(self._h, {"weights": 0, "grad": 1, "s": 2, "r": 3}[what], out.ctypes.data_as(C.c_void_p), n))
        return out

    # ---- Deep-Retrieval (row A13)
    def dr_load_model(self, weights, E, L, K, D, num_item, dtype=np.float64):
        """weights: dict with layer_emb, layer_w [D], layer_b [D] and optionally rerank_emb, rerank_w, rerank_b,
        softmax_w, softmax_b (numpy arrays; converted to `dtype`, which is also the arithmetic type)."""
        dt = np.dtype(dtype)
        f = lambda a: np.ascontiguousarray(a, dtype=dt)
        keep = dict(layer_emb=f(weights["layer_emb"]), layer_w=[f(w) for w in weights["layer_w"]],
                    layer_b=[f(b) for b in weights["layer_b"]])
        assert keep["layer_emb"].size == (num_item + K * (D - 1)) * E
        for d in range(D):
            assert keep["layer_w"][d].size == K * (L + d) * E and keep["layer_b"][d].size == K
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        m = N.DrModel()
        m.dtype = 0 if dt == np.float32 else 1
        m.on_device = 0
        m.embed, m.seq_len, m.num_node, m.num_layer, m.num_item = E, L, K, D, num_item
        m.layer_emb = vp(keep["layer_emb"])
        wp = (C.c_void_p * D)(*[vp(w) for w in keep["layer_w"]])
        bp = (C.c_void_p * D)(*[vp(b) for b in keep["layer_b"]])
        m.layer_w, m.layer_b = wp, bp
        if weights.get("rerank_emb") is not None:
            for k in ("rerank_emb", "rerank_w", "rerank_b", "softmax_w", "softmax_b"):
                keep[k] = f(weights[k])
                setattr(m, k, vp(keep[k]))
        self._chk(N.lib().dm_dr_load_model(self._h, C.byref(m)))
        self.dr_dims = dict(E=E, L=L, K=K, D=D, num_item=num_item, dtype=dt)

    def dr_load_model_dev(self, ptrs, E, L, K, D, num_item, dtype=np.float32):
        """like dr_load_model but every entry of `ptrs` is a device pointer (c_void_p); layer_w / layer_b are lists."""
        dt = np.dtype(dtype)
        m = N.DrModel()
        m.dtype = 0 if dt == np.float32 else 1
        m.on_device = 1
        m.embed, m.seq_len, m.num_node, m.num_layer, m.num_item = E, L, K, D, num_item
        m.layer_emb = ptrs["layer_emb"]
        wp = (C.c_void_p * D)(*ptrs["layer_w"])
        bp = (C.c_void_p * D)(*ptrs["layer_b"])
        m.layer_w, m.layer_b = wp, bp
        if ptrs.get("rerank_emb") is not None:
            for k in ("rerank_emb", "rerank_w", "rerank_b", "softmax_w", "softmax_b"):
                setattr(m, k, ptrs[k])
        self._chk(N.lib().dm_dr_load_model(self._h, C.byref(m)))
        self.dr_dims = dict(E=E, L=L, K=K, D=D, num_item=num_item, dtype=dt)

    def dr_load_model_synthetic(self, E, L, K, D, num_item, seed, scale=0.05, rerank=True, dtype=np.float32):
        """Random-init Deep-Retrieval model generated ON THE DEVICE (N(0, scale) matrices, zero biases — the reference's
        init, RerankModel.scala:15-16) and loaded without a host copy (bench only).  dtype float64 (the reference's
        arithmetic type) holds the same draws as float32, widened."""
        dt = np.dtype(dtype)
        fill = lambda n, sd, std: self._fill_new(n, sd, std, dt)
        ptrs = dict(layer_emb=fill((num_item + K * (D - 1)) * E, seed + 1, scale),
                    layer_w=[fill(K * (L + d) * E, seed + 10 + d, scale) for d in range(D)],
                    layer_b=[fill(K, seed + 20 + d, 0.0) for d in range(D)])
        if rerank:
            ptrs.update(rerank_emb=fill(num_item * E, seed + 2, scale), rerank_w=fill(E * L * E, seed + 3, scale),
                        rerank_b=fill(E, seed + 4, 0.0), softmax_w=fill(num_item * E, seed + 5, scale),
                        softmax_b=fill(num_item, seed + 6, 0.0))
        self.dr_load_model_dev(ptrs, E, L, K, D, num_item, dtype=dt)
        self.synchronize()
        for k, v in ptrs.items():
            for q in (v if isinstance(v, list) else [v]):
                self.dev_free(q)

    def _fill_new(self, n, seed, std, dtype=np.float32):
        dt = np.dtype(dtype)
        d = self.dev_alloc(int(n) * dt.itemsize)
        fn = N.lib().dm_fill_normal if dt == np.float32 else N.lib().dm_fill_normal_f64
        self._chk(fn(self._h, d, int(n), 0.0, float(std), int(seed)))
        return d

    def dr_load_path_items(self, path_nodes, item_off, items):
        pn = _i32(path_nodes).reshape(-1, self.dr_dims["D"])
        off = np.ascontiguousarray(item_off, dtype=np.int64)
        it = _i32(items)
        assert off.size == len(pn) + 1
        self._chk(N.lib().dm_dr_load_path_items(self._h, _p(pn, N.i32p), len(pn), _p(off, N.i64p), _p(it, N.i32p)))

    def dr_beam_search(self, seq_ids, beam):
        seq = _i32(seq_ids)
        if seq.ndim == 1:
            seq = seq[None, :]
        U, D = seq.shape[0], self.dr_dims["D"]
        assert seq.shape[1] == self.dr_dims["L"]
        paths = np.empty((U, beam, D), np.int32)
        probs = np.empty((U, beam), np.float64)
        cnt = np.empty(U, np.int32)
        self._chk(N.lib().dm_dr_beam_search(self._h, _p(seq, N.i32p), U, int(beam), _p(paths, N.i32p),
                                            probs.ctypes.data_as(C.POINTER(C.c_double)), _p(cnt, N.i32p)))
        return paths, probs, cnt

    def dr_recommend(self, seq_ids, beam, topk):
        seq = _i32(seq_ids)
        if seq.ndim == 1:
            seq = seq[None, :]
        U = seq.shape[0]
        assert seq.shape[1] == self.dr_dims["L"]
        ids = np.empty((U, topk), np.int32)
        sc = np.empty((U, topk), np.float64)
        cnt = np.empty(U, np.int32)
        self._chk(N.lib().dm_dr_recommend(self._h, _p(seq, N.i32p), U, int(beam), int(topk), _p(ids, N.i32p),
                                          sc.ctypes.data_as(C.POINTER(C.c_double)), _p(cnt, N.i32p)))
        return ids, sc, cnt

    def dr_beam_search_dev(self, d_seq, U, beam, d_paths, d_probs, d_counts):
        self._chk(N.lib().dm_dr_beam_search_dev(self._h, d_seq, U, int(beam), d_paths, d_probs, d_counts))

    def dr_recommend_dev(self, d_seq, U, beam, topk, d_ids, d_scores, d_counts):
        self._chk(N.lib().dm_dr_recommend_dev(self._h, d_seq, U, int(beam), int(topk), d_ids, d_scores, d_counts))

    # ---- device-resident path (bench)
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(N.lib().dm_dev_alloc(self._h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        self._chk(N.lib().dm_dev_free(self._h, p))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(N.lib().dm_memcpy_h2d(self._h, dptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def d2h(self, arr, dptr):
        self._chk(N.lib().dm_memcpy_d2h(self._h, arr.ctypes.data_as(C.c_void_p), dptr, arr.nbytes))

    def tdm_beam_search_dev(self, d_seq, U, L, beam, topk, d_ids, d_scores, d_counts, use_mask=True):
        opts = N.SearchOpts(int(beam), int(topk), int(bool(use_mask)), 0)
        self._chk(N.lib().dm_tdm_beam_search_dev(self._h, d_seq, U, L, C.byref(opts), None, None, d_ids, d_scores,
                                                 d_counts))

    def otm_beam_search_dev(self, d_seq, U, L, beam, leaf_level, d_ids, d_scores, d_counts):
        self._chk(N.lib().dm_otm_beam_search_dev(self._h, d_seq, U, L, int(beam), int(leaf_level), d_ids, d_scores, d_counts))

    def synchronize(self):
        self._chk(N.lib().dm_synchronize(self._h))

    _SCORER = {"f32": 0, "split_f16": 1, "auto": 2, "f64": 3}

    def set_scorer_mode(self, mode):
        """Arithmetic of the beam-search scorer: "f32" (fp32-input MFMA), "split_f16" (fp16 hi/lo operand split on the
        fp16 matrix pipe, fp32 accumulation) or "auto" (default: split_f16 where E allows); include/dismember_hip.h."""
        self._chk(N.lib().dm_set_scorer_mode(self._h, int(self._SCORER.get(mode, mode))))

    def scorer_mode(self):
        m, eff, se, sw = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(N.lib().dm_get_scorer_mode(self._h, C.byref(m), C.byref(eff), C.byref(se), C.byref(sw)))
        names = {v: k for k, v in self._SCORER.items()}
        return {"mode": names[eff.value], "setting": names[m.value], "shift_emb": se.value, "shift_w": sw.value}

    def timing_reset(self):
        self._chk(N.lib().dm_kernel_timing_reset(self._h))

    def timing_get(self):
        n, ms = C.c_int(0), C.c_double(0)
        self._chk(N.lib().dm_kernel_timing_get(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def timing_get_kind(self, kind):
        """launches and summed milliseconds of one kind of launch (0 = search kernels, 1 = deferred-user second pass)"""
        n, ms = C.c_int(0), C.c_double(0)
        self._chk(N.lib().dm_kernel_timing_get_kind(self._h, int(kind), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def last_beam_kernel(self):
        return (N.lib().dm_last_beam_kernel(self._h) or b"").decode()

    def last_scored_rows(self):
        r = C.c_int64(0)
        self._chk(N.lib().dm_last_scored_rows(self._h, C.byref(r)))
        return r.value
