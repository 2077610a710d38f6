"""TEST INFRASTRUCTURE: ctypes bindings of the CPU oracle (oracle/libdm_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It never touches the HIP library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libdm_oracle.so")

i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)

SCORER_F32 = C.CFUNCTYPE(C.c_int, C.c_void_p, i32p, C.c_int, i32p, C.c_int, i32p, C.c_int, f32p)
SCORER_F64 = C.CFUNCTYPE(C.c_int, C.c_void_p, i32p, C.c_int, i32p, C.c_int, f64p)


def build(force=False):
    src = [os.path.join(_DIR, f) for f in ("dm_oracle.c", "din_body.inc", "dr_body.inc", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_din_create_f32.restype = C.c_void_p
        L.orc_din_create_f32.argtypes = [C.c_int, C.c_int, C.c_int64, f32p, C.c_int64]
        L.orc_din_create_f64.restype = C.c_void_p
        L.orc_din_create_f64.argtypes = [C.c_int, C.c_int, C.c_int64, f64p, C.c_int64]
        L.orc_din_destroy_f32.argtypes = [C.c_void_p]
        L.orc_din_destroy_f64.argtypes = [C.c_void_p]
        L.orc_din_forward_f32.argtypes = [C.c_void_p, i32p, i32p, i32p, C.c_int64, C.c_int64, f32p]
        L.orc_din_forward_f64.argtypes = [C.c_void_p, i32p, i32p, i32p, C.c_int64, C.c_int64, f64p]
        L.orc_din_train_grads_f32.restype = C.c_float
        L.orc_din_train_grads_f32.argtypes = [C.c_void_p, i32p, i32p, i32p, C.c_int64, f32p, C.c_int64, f32p]
        L.orc_din_train_grads_f64.restype = C.c_double
        L.orc_din_train_grads_f64.argtypes = [C.c_void_p, i32p, i32p, i32p, C.c_int64, f64p, C.c_int64, f64p]
        L.orc_adam_step_f32.argtypes = [f32p, f32p, f32p, f32p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.POINTER(C.c_int)]
        L.orc_adam_step_f64.argtypes = [f64p, f64p, f64p, f64p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.POINTER(C.c_int)]
        L.orc_softmax_f32.argtypes = [f32p, f32p, C.c_int, C.c_int]
        L.orc_softmax_f64.argtypes = [f64p, f64p, C.c_int, C.c_int]
        L.orc_softmax_backward_f64.argtypes = [f64p, f64p, f64p, C.c_int, C.c_int]
        L.orc_softmax_backward_f32.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int]
        L.orc_java_float_compare.argtypes = [C.c_float, C.c_float]
        L.orc_stable_argsort_desc_f32.argtypes = [f32p, i32p, C.c_int]
        L.orc_stable_argsort_desc_f64.argtypes = [f64p, i32p, C.c_int]
        L.orc_tree_create.restype = C.c_void_p
        L.orc_tree_create.argtypes = [i32p, i32p, u8p, C.c_int64, i32p, i32p, C.c_int64, C.c_int]
        L.orc_tree_destroy.argtypes = [C.c_void_p]
        L.orc_tree_non_leaf_offset.argtypes = [C.c_void_p]
        L.orc_tree_max_code.argtypes = [C.c_void_p]
        L.orc_tdm_id_to_code.argtypes = [C.c_void_p, i32p, C.c_int, i32p, i32p]
        L.orc_level_start.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_tdm_level_step.argtypes = [C.c_void_p, C.c_int, i32p, f32p, C.c_int, i32p, f32p,
                                         C.POINTER(C.c_int), i32p]
        L.orc_tdm_finalize.argtypes = [C.c_void_p, i32p, f32p, C.c_int, i32p, C.c_int, C.c_int, i32p, f32p]
        L.orc_tdm_recommend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, i32p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, i32p, C.c_int, i32p, f32p, i32p, f32p, i32p,
                                        C.POINTER(C.c_int)]
        L.orc_tdm_recommend_items.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, i32p, C.c_int, C.c_int,
                                              C.c_int, C.c_int, i32p, C.c_int, C.c_int, i32p, f32p]
        L.orc_tdm_recommend_batch.argtypes = [C.c_void_p, C.c_void_p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, i32p, f32p, i32p]
        L.orc_jtm_child_weights.argtypes = [C.c_void_p, C.c_void_p, i32p, i64p, i32p, i32p, C.c_int64, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_jtm_rebalance.argtypes = [i32p, f32p, i32p, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_int, i32p]
        L.orc_jtm_id_to_code_with_mask.argtypes = [C.c_void_p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p]
        L.orc_lower_log2.argtypes = [C.c_int]
        L.orc_upper_log2.argtypes = [C.c_int]
        L.orc_otm_beam_nodes.argtypes = [i32p, f64p, C.c_int, C.c_int, C.c_int, i32p]
        L.orc_otm_beam_search.argtypes = [C.c_void_p, C.c_void_p, i32p, C.c_int, C.c_int, C.c_int, i32p, f64p]
        L.orc_otm_finalize.argtypes = [i32p, f64p, C.c_int, i32p, C.c_int64, C.c_int, i32p, f64p]
        pp = C.POINTER(f64p)
        L.orc_dr_create.restype = C.c_void_p
        L.orc_dr_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, f64p, pp, pp, f64p, f64p, f64p, f64p, f64p]
        L.orc_dr_destroy.argtypes = [C.c_void_p]
        L.orc_dr_inference.argtypes = [C.c_void_p, i32p, C.c_int, C.c_int, f64p]
        L.orc_dr_softmax.argtypes = [f64p, C.c_int, f64p]
        L.orc_dr_beam_search.argtypes = [C.c_void_p, i32p, C.c_int, i32p, f64p]
        L.orc_dr_search_candidates.restype = C.c_int64
        L.orc_dr_search_candidates.argtypes = [i32p, C.c_int, C.c_int, i32p, C.c_int64, i64p, i32p, i32p, C.c_int64]
        L.orc_dr_rerank.argtypes = [C.c_void_p, i32p, C.c_int64, i32p, f64p]
        L.orc_dr_recommend.argtypes = [C.c_void_p, i32p, C.c_int, C.c_int, i32p, C.c_int64, i64p, i32p, i32p, f64p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Din:
    """DIN restatement holding the compact A0 parameter vector."""

    def __init__(self, weights, E, L, num_index):
        self.dtype = np.dtype(weights.dtype)
        assert self.dtype in (np.float32, np.float64)
        self.w = np.ascontiguousarray(weights)
        self.E, self.L, self.num_index = E, L, num_index
        self.sfx = "f32" if self.dtype == np.float32 else "f64"
        self.ptr_t = f32p if self.dtype == np.float32 else f64p
        self.h = getattr(lib(), "orc_din_create_" + self.sfx)(E, L, num_index, _p(self.w, self.ptr_t), self.w.size)
        if not self.h:
            raise ValueError("weight vector length does not match (E, num_index)")

    def __del__(self):
        if getattr(self, "h", None):
            getattr(lib(), "orc_din_destroy_" + self.sfx)(self.h)
            self.h = None

    def forward(self, codes, seqs, pad_flat=None):
        codes = _i32(codes).ravel()
        seqs = _i32(seqs).reshape(-1)
        B = codes.size
        assert seqs.size == B * self.L
        pad = _i32([] if pad_flat is None else pad_flat).ravel()
        out = np.empty(B, dtype=self.dtype)
        rc = getattr(lib(), "orc_din_forward_" + self.sfx)(self.h, _p(codes, i32p), _p(seqs, i32p), _p(pad, i32p),
                                                            pad.size, B, _p(out, self.ptr_t))
        if rc != 0:
            raise IndexError("embeddingLookup failed at row %d" % (-rc - 1))
        return out

    def train_grads(self, codes, seqs, pad_flat, labels, grad=None):
        """One worker's trainBatch: returns (mean BCE loss, gradient in the compact-vector layout)."""
        codes = _i32(codes).ravel()
        seqs = _i32(seqs).reshape(-1)
        pad = _i32([] if pad_flat is None else pad_flat).ravel()
        lab = np.ascontiguousarray(labels, dtype=self.dtype).ravel()
        if grad is None:
            grad = np.zeros(self.w.size, self.dtype)
        fn = getattr(lib(), "orc_din_train_grads_" + self.sfx)
        loss = fn(self.h, _p(codes, i32p), _p(seqs, i32p), _p(pad, i32p), pad.size, _p(lab, self.ptr_t), codes.size,
                  _p(grad, self.ptr_t))
        return float(loss), grad

    @property
    def scorer_f32(self):
        return C.cast(lib().orc_din_scorer_f32, C.c_void_p)

    @property
    def scorer_f64(self):
        return C.cast(lib().orc_din_scorer_f64, C.c_void_p)


class TdmTree:
    def __init__(self, codes, ids, is_leaf, leaf_ids, leaf_codes, max_level):
        self.codes, self.ids = _i32(codes), _i32(ids)
        self.is_leaf = np.ascontiguousarray(is_leaf, dtype=np.uint8)
        self.leaf_ids, self.leaf_codes = _i32(leaf_ids), _i32(leaf_codes)
        self.max_level = int(max_level)
        self.h = lib().orc_tree_create(_p(self.codes, i32p), _p(self.ids, i32p), _p(self.is_leaf, u8p),
                                       self.codes.size, _p(self.leaf_ids, i32p), _p(self.leaf_codes, i32p),
                                       self.leaf_ids.size, self.max_level)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tree_destroy(self.h)
            self.h = None

    @property
    def non_leaf_offset(self):
        return lib().orc_tree_non_leaf_offset(self.h)

    @property
    def max_code(self):
        return lib().orc_tree_max_code(self.h)

    def id_to_code(self, item_ids):
        ids = _i32(item_ids)
        codes = np.empty_like(ids)
        mask = np.empty_like(ids)
        nm = lib().orc_tdm_id_to_code(self.h, _p(ids, i32p), ids.size, _p(codes, i32p), _p(mask, i32p))
        return codes, mask[:nm].copy()

    def level_step(self, beam, cand_codes, cand_preds):
        cc = _i32(cand_codes)
        cp = np.ascontiguousarray(cand_preds, dtype=np.float32)
        n = cc.size
        lc = np.empty(max(n, 1), np.int32)
        lp = np.empty(max(n, 1), np.float32)
        ch = np.empty(max(2 * n, 2), np.int32)
        nl = C.c_int(0)
        nc = lib().orc_tdm_level_step(self.h, beam, _p(cc, i32p), _p(cp, f32p), n, _p(lc, i32p), _p(lp, f32p),
                                      C.byref(nl), _p(ch, i32p))
        return lc[:nl.value].copy(), lp[:nl.value].copy(), ch[:nc].copy()

    def finalize(self, leaf_codes, leaf_preds, topk, consumed=()):
        lc = _i32(leaf_codes)
        lp = np.ascontiguousarray(leaf_preds, dtype=np.float32)
        cs = _i32(list(consumed))
        oi = np.empty(max(topk, 1), np.int32)
        op = np.empty(max(topk, 1), np.float32)
        k = lib().orc_tdm_finalize(self.h, _p(lc, i32p), _p(lp, f32p), lc.size, _p(cs, i32p), cs.size, topk,
                                   _p(oi, i32p), _p(op, f32p))
        return oi[:k].copy(), op[:k].copy()

    def recommend(self, din, seq_ids, topk, beam, use_mask=True, consumed=(), trace=False, scorer=None, ctx=None):
        """TDM.recommend (logits, not sigmoid). scorer: optional SCORER_F32 python callback."""
        seq = _i32(seq_ids)
        cs = _i32(list(consumed))
        oi = np.empty(max(topk, 1), np.int32)
        op = np.empty(max(topk, 1), np.float32)
        if scorer is None:
            fn, cx = din.scorer_f32, din.h
        else:
            fn, cx = C.cast(scorer, C.c_void_p), ctx
        if trace:
            nlev = self.max_level + 2
            tc = np.empty(2 * beam * nlev, np.int32)
            tp = np.empty(2 * beam * nlev, np.float32)
            tn = np.zeros(nlev, np.int32)
            tl = C.c_int(0)
            k = lib().orc_tdm_recommend(self.h, fn, cx, _p(seq, i32p), seq.size, topk, beam, int(use_mask),
                                        _p(cs, i32p), cs.size, _p(oi, i32p), _p(op, f32p), _p(tc, i32p),
                                        _p(tp, f32p), _p(tn, i32p), C.byref(tl))
        else:
            k = lib().orc_tdm_recommend(self.h, fn, cx, _p(seq, i32p), seq.size, topk, beam, int(use_mask),
                                        _p(cs, i32p), cs.size, _p(oi, i32p), _p(op, f32p), None, None, None, None)
        if k < 0:
            raise RuntimeError("oracle scorer failed: %d" % k)
        if trace:
            levels, off = [], 0
            for i in range(tl.value):
                n = int(tn[i])
                levels.append((tc[off:off + n].copy(), tp[off:off + n].copy()))
                off += n
            return oi[:k].copy(), op[:k].copy(), levels
        return oi[:k].copy(), op[:k].copy()

    def recommend_batch(self, din, seqs, topk, beam, use_mask=True, n_threads=1):
        """TDM.recommend for a [U, L] batch on n_threads worker threads (contiguous user ranges)."""
        seqs = _i32(seqs)
        U, L = seqs.shape
        oi = np.full((U, topk), -1, np.int32)
        op = np.zeros((U, topk), np.float32)
        oc = np.zeros(U, np.int32)
        lib().orc_tdm_recommend_batch(self.h, din.h, _p(seqs, i32p), U, L, topk, beam, int(use_mask), int(n_threads),
                                      _p(oi, i32p), _p(op, f32p), _p(oc, i32p))
        return oi, op, oc

    def recommend_items(self, din, seq_ids, topk, beam, use_mask=True, consumed=None):
        seq = _i32(seq_ids)
        cs = _i32([] if consumed is None else list(consumed))
        oi = np.empty(max(topk, 1), np.int32)
        op = np.empty(max(topk, 1), np.float32)
        k = lib().orc_tdm_recommend_items(self.h, din.scorer_f32, din.h, _p(seq, i32p), seq.size, topk, beam,
                                          int(use_mask), _p(cs, i32p), cs.size, int(consumed is not None),
                                          _p(oi, i32p), _p(op, f32p))
        if k < 0:
            raise RuntimeError("oracle scorer failed: %d" % k)
        return oi[:k].copy()


def tdm_sample_batch(tree, seq_item_ids, target_item_ids, neg_counts, start_level=1, seed=0, use_mask=True, with_prob=False,
                     tolerance=20, node_codes=None, node_probs=None):
    """NegativeSampler.sample + MiniBatch.convert with the product's counter-based stream: (codes, seqs, rowmask, labels)."""
    seq = _i32(seq_item_ids)
    tgt = _i32(target_item_ids).ravel()
    T, L = seq.shape
    neg = _i32(neg_counts)
    per = int(sum(1 + int(neg[l]) for l in range(start_level, tree.max_level + 1)))
    cap = max(T * per, 1)
    codes = np.empty(cap, np.int32); seqs = np.empty((cap, L), np.int32); mask = np.empty(cap, np.uint32); lab = np.empty(cap, np.float32)
    nc = _i32([] if node_codes is None else node_codes)
    npb = np.ascontiguousarray([] if node_probs is None else node_probs, dtype=np.float32)
    fn = lib().orc_tdm_sample_batch
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, i32p, i32p, C.c_int64, C.c_int, i32p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, i32p, f32p, C.c_int64,
                   i32p, i32p, C.POINTER(C.c_uint32), f32p]
    n = fn(tree.h, _p(seq, i32p), _p(tgt, i32p), T, L, _p(neg, i32p), int(start_level), int(seed), int(bool(use_mask)),
           int(bool(with_prob)), int(tolerance), _p(nc, i32p), _p(npb, f32p), nc.size, _p(codes, i32p), _p(seqs, i32p),
           mask.ctypes.data_as(C.POINTER(C.c_uint32)), _p(lab, f32p))
    return codes[:n].copy(), seqs[:n].copy(), mask[:n].copy(), lab[:n].copy()


def level_start(n):
    s, l = C.c_int(0), C.c_int(0)
    lib().orc_level_start(n, C.byref(s), C.byref(l))
    return s.value, l.value


def otm_beam_search(din, seq_codes, leaf_level, beam):
    """CandidateSearcher.beamSearch with the f64 DIN restatement."""
    seq = _i32(seq_codes)
    cap = 4 * beam + 4
    ids = np.empty(cap, np.int32)
    sc = np.empty(cap, np.float64)
    n = lib().orc_otm_beam_search(din.scorer_f64, din.h, _p(seq, i32p), seq.size, leaf_level, beam, _p(ids, i32p),
                                  _p(sc, f64p))
    if n < 0:
        raise RuntimeError("oracle scorer failed: %d" % n)
    return ids[:n].copy(), sc[:n].copy()


def otm_finalize(ids, scores, node_to_item, topk):
    ids = _i32(ids)
    sc = np.ascontiguousarray(scores, dtype=np.float64)
    n2i = _i32(node_to_item)
    oi = np.empty(max(topk, 1), np.int32)
    os_ = np.empty(max(topk, 1), np.float64)
    k = lib().orc_otm_finalize(_p(ids, i32p), _p(sc, f64p), ids.size, _p(n2i, i32p), n2i.size, topk, _p(oi, i32p),
                               _p(os_, f64p))
    return oi[:k].copy(), os_[:k].copy()


def softmax(x):
    x = np.ascontiguousarray(x)
    out = np.empty_like(x)
    n, dim = int(np.prod(x.shape[:-1])), x.shape[-1]
    if x.dtype == np.float64:
        lib().orc_softmax_f64(_p(x, f64p), _p(out, f64p), n, dim)
    else:
        lib().orc_softmax_f32(_p(x, f32p), _p(out, f32p), n, dim)
    return out


def softmax_backward(out, gout):
    out = np.ascontiguousarray(out, dtype=np.float64)
    gout = np.ascontiguousarray(gout, dtype=np.float64)
    gin = np.empty_like(out)
    lib().orc_softmax_backward_f64(_p(out, f64p), _p(gout, f64p), _p(gin, f64p), int(np.prod(out.shape[:-1])),
                                   out.shape[-1])
    return gin


i64p = C.POINTER(C.c_int64)


def jtm_child_weights(tree, din, items, row_off, row_ids, item_node, L, old_level, level, hierarchical=False,
                      min_level=0, use_mask=True):
    items, item_node, row_ids = _i32(items), _i32(item_node), _i32(row_ids)
    row_off = np.ascontiguousarray(row_off, np.int64)
    n = items.size
    w = np.empty((n, 1 << (level - old_level)), np.float32)
    lib().orc_jtm_child_weights(tree.h, din.h, _p(items, i32p), _p(row_off, i64p), _p(row_ids, i32p), _p(item_node, i32p),
                                n, L, old_level, level, int(hierarchical), min_level, int(use_mask), _p(w, f32p))
    return w


def jtm_rebalance(items, weights, old_node, node, old_level, level, max_assign):
    items, old_node = _i32(items), _i32(old_node)
    weights = np.ascontiguousarray(weights, np.float32)
    out = np.empty(items.size, np.int32)
    lib().orc_jtm_rebalance(_p(items, i32p), _p(weights, f32p), _p(old_node, i32p), items.size, int(node), old_level,
                            level, int(max_assign), _p(out, i32p))
    return out


class Adam:
    """Adam.optimize (scalann/.../optim/Adam.scala:19-73) on a flat numpy vector, state kept here."""

    def __init__(self, n, dtype=np.float32, lr=1e-3, lrd=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
        self.s = np.zeros(n, dtype)
        self.r = np.zeros(n, dtype)
        self.t = C.c_int(0)
        self.hp = (lr, lrd, beta1, beta2, eps)
        self.dtype = np.dtype(dtype)

    def step(self, w, g):
        pt = f32p if self.dtype == np.float32 else f64p
        fn = lib().orc_adam_step_f32 if self.dtype == np.float32 else lib().orc_adam_step_f64
        fn(_p(w, pt), _p(g, pt), _p(self.s, pt), _p(self.r, pt), w.size, *self.hp, C.byref(self.t))


class DeepRetrieval:
    """fp64 restatement of the Deep-Retrieval serving path (oracle/dr_body.inc).

    weights: dict(layer_emb [(num_item + K(D-1)) x E], layer_w [D] of [K x (L+d)E], layer_b [D] of [K],
    rerank_emb [num_item x E], rerank_w [E x L*E], rerank_b [E], softmax_w [num_item x E], softmax_b [num_item]).
    path_items: optional (path_nodes [P x D], item_off [P+1], items) -- sorted here.
    """

    def __init__(self, weights, E, L, K, D, num_item, path_items=None):
        L_ = lib()
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.E, self.L, self.K, self.D, self.num_item = E, L, K, D, num_item
        self._keep = dict(layer_emb=f(weights["layer_emb"]), layer_w=[f(w) for w in weights["layer_w"]],
                          layer_b=[f(b) for b in weights["layer_b"]], rerank_emb=f(weights["rerank_emb"]),
                          rerank_w=f(weights["rerank_w"]), rerank_b=f(weights["rerank_b"]),
                          softmax_w=f(weights["softmax_w"]), softmax_b=f(weights["softmax_b"]))
        k = self._keep
        wp = (f64p * D)(*[_p(w, f64p) for w in k["layer_w"]])
        bp = (f64p * D)(*[_p(b, f64p) for b in k["layer_b"]])
        self.h = L_.orc_dr_create(E, L, K, D, num_item, _p(k["layer_emb"], f64p), wp, bp, _p(k["rerank_emb"], f64p),
                                  _p(k["rerank_w"], f64p), _p(k["rerank_b"], f64p), _p(k["softmax_w"], f64p),
                                  _p(k["softmax_b"], f64p))
        self.path_items = None
        if path_items is not None:
            self.set_path_items(*path_items)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_dr_destroy(self.h)
            self.h = None

    def set_path_items(self, path_nodes, item_off, items):
        pn = np.asarray(path_nodes, np.int32).reshape(-1, self.D)
        off = np.asarray(item_off, np.int64)
        items = np.asarray(items, np.int32)
        order = np.lexsort(pn.T[::-1])
        cnt = (off[1:] - off[:-1])[order]
        new_off = np.zeros(len(pn) + 1, np.int64)
        np.cumsum(cnt, out=new_off[1:])
        new_items = np.concatenate([items[off[o]:off[o + 1]] for o in order]) if len(order) else items[:0]
        self.path_items = (np.ascontiguousarray(pn[order]), new_off, np.ascontiguousarray(new_items, dtype=np.int32))

    def inference(self, input_ids, rank):
        x = _i32(input_ids)
        out = np.empty(self.K, np.float64)
        lib().orc_dr_inference(self.h, _p(x, i32p), len(x), rank, _p(out, f64p))
        return out

    def beam_search(self, seq, beam):
        s = _i32(seq)
        assert len(s) == self.L
        paths = np.empty((max(beam, 1), self.D), np.int32)
        probs = np.empty(max(beam, 1), np.float64)
        n = lib().orc_dr_beam_search(self.h, _p(s, i32p), beam, _p(paths, i32p), _p(probs, f64p))
        return paths[:n].copy(), probs[:n].copy()

    def search_candidates(self, paths):
        pn, off, items = self.path_items
        paths = np.ascontiguousarray(paths, np.int32)
        n = lib().orc_dr_search_candidates(_p(paths, i32p), len(paths), self.D, _p(pn, i32p), len(pn), _p(off, i64p),
                                           _p(items, i32p), None, 0)
        out = np.empty(max(n, 1), np.int32)
        lib().orc_dr_search_candidates(_p(paths, i32p), len(paths), self.D, _p(pn, i32p), len(pn), _p(off, i64p),
                                       _p(items, i32p), _p(out, i32p), n)
        return out[:n]

    def rerank(self, cands, seq):
        c = _i32(cands)
        s = _i32(seq)
        out = np.empty(max(len(c), 1), np.float64)
        lib().orc_dr_rerank(self.h, _p(c, i32p), len(c), _p(s, i32p), _p(out, f64p))
        return out[:len(c)]

    def recommend(self, seq, topk, beam):
        pn, off, items = self.path_items
        s = _i32(seq)
        ids = np.empty(max(topk, 1), np.int32)
        sc = np.empty(max(topk, 1), np.float64)
        n = lib().orc_dr_recommend(self.h, _p(s, i32p), topk, beam, _p(pn, i32p), len(pn), _p(off, i64p), _p(items, i32p),
                                   _p(ids, i32p), _p(sc, f64p))
        return ids[:n].copy(), sc[:n].copy()


def dr_softmax(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib().orc_dr_softmax(_p(x, f64p), len(x), _p(out, f64p))
    return out
