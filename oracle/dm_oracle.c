/*
 * TEST INFRASTRUCTURE — not product code.
 *
 * CPU restatement ("oracle") of dismember's TDM / OTM beam-search retrieval
 * path and its DIN scorer, in plain C.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (dismember_amd/csrc) never does.
 *
 * PARITY UNPINNED: the reference (Scala 2.13 + JVM + MKL JNI) can be neither
 * compiled nor run in the build container and none of its own tests pins a
 * score, a beam or a tree assignment.  What *is* pinned:
 *   - SoftMax forward/backward known answers (scalann/src/test/scala/SoftMaxTest.scala:13,23)
 *   - structural invariants restated in tests/test_oracle.py
 *   - the reference's bundled trained weights / tree (tests/golden/ .npy and .npz files)
 *     as golden INPUTS; outputs on them are restatement-derived.
 *   - what the bundled TRAINED weights separate when read through this restatement (round 5,
 *     tests/test_oracle.py::test_trained_weights_pin_layout, tools/trained_weights_pin_probe.py): the concat order
 *     [item; att] of linear1's input, the positions of l1.b / l2.W / l2.b in the compact vector and the sign of l2.W —
 *     NOT the orientation of l1.W / att.W (the bundled models are not converged: +-0.01 BCE either way).
 *   - the OTM path end to end on the reference's artefacts (round 5, tests/test_oracle.py::
 *     test_otm_trained_model_pins_mapping_search_and_evaluator): the bundled trained DIN[Double] with the bundled item -> node
 *     mapping explains the bundled interactions (recall@10 0.0144, eval loss 3.02) against 0.0008 .. 0.0038 / 3.65 .. 3.86 when
 *     the same nodes are dealt to the items at random — the id space of the mapping file, the leaf range, the fp64 forward's
 *     layout, the beam search and the evaluator's consumed / allNodes filter are all on that path.
 *
 * Citation prefixes:  T/ = tdm/src/main/scala/com/mass/tdm/
 *                     O/ = otm/src/main/scala/com/mass/otm/
 *                     S/ = scalann/src/main/scala/com/mass/scalann/
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ DIN */
#define REAL float
#define SUFFIX f32
#define REAL_EXP expf
#define REAL_LOG logf
#define REAL_SQRT sqrtf
#include "din_body.inc"
#undef REAL
#undef SUFFIX
#undef REAL_EXP
#undef REAL_LOG
#undef REAL_SQRT

#define REAL double
#define SUFFIX f64
#define REAL_EXP exp
#define REAL_LOG log
#define REAL_SQRT sqrt
#include "din_body.inc"
#undef REAL
#undef SUFFIX
#undef REAL_EXP
#undef REAL_LOG
#undef REAL_SQRT

/* --------------------------------------------------- Java number orderings */

/* java.lang.Float.compare / compareTo: total order, -0.0 < 0.0, NaN greatest, all NaN equal */
static int java_float_compare(float x, float y) {
  if (x < y) return -1;
  if (x > y) return 1;
  int32_t a, b;
  if (x != x) a = 0x7fc00000; else memcpy(&a, &x, 4); /* floatToIntBits canonicalises NaN */
  if (y != y) b = 0x7fc00000; else memcpy(&b, &y, 4);
  return a == b ? 0 : (a < b ? -1 : 1);
}
static int java_double_compare(double x, double y) {
  if (x < y) return -1;
  if (x > y) return 1;
  int64_t a, b;
  if (x != x) a = 0x7ff8000000000000LL; else memcpy(&a, &x, 8);
  if (y != y) b = 0x7ff8000000000000LL; else memcpy(&b, &y, 8);
  return a == b ? 0 : (a < b ? -1 : 1);
}
int orc_java_float_compare(float x, float y) { return java_float_compare(x, y); }

/*
 * Stable descending argsort (JVM Arrays.sort(Object[]) / sortBy are stable
 * merges; comparator y.pred.compareTo(x.pred), T/model/Recommender.scala:77-84).
 * Plain insertion-merge: bottom-up stable merge sort on an index array.
 */
static void stable_argsort_desc_f32(const float *v, int32_t *idx, int n) {
  int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  for (int i = 0; i < n; i++) idx[i] = i;
  for (int w = 1; w < n; w *= 2) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int a = lo, b = mid, o = lo;
      while (a < mid && b < hi) {
        /* take from the right run only if it is strictly "smaller" under the descending comparator */
        if (java_float_compare(v[idx[a]], v[idx[b]]) < 0) tmp[o++] = idx[b++];
        else tmp[o++] = idx[a++];
      }
      while (a < mid) tmp[o++] = idx[a++];
      while (b < hi) tmp[o++] = idx[b++];
    }
    memcpy(idx, tmp, sizeof(int32_t) * n);
  }
  free(tmp);
}
static void stable_argsort_desc_f64(const double *v, int32_t *idx, int n) {
  int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  for (int i = 0; i < n; i++) idx[i] = i;
  for (int w = 1; w < n; w *= 2) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int a = lo, b = mid, o = lo;
      while (a < mid && b < hi) {
        if (java_double_compare(v[idx[a]], v[idx[b]]) < 0) tmp[o++] = idx[b++];
        else tmp[o++] = idx[a++];
      }
      while (a < mid) tmp[o++] = idx[a++];
      while (b < hi) tmp[o++] = idx[b++];
    }
    memcpy(idx, tmp, sizeof(int32_t) * n);
  }
  free(tmp);
}
void orc_stable_argsort_desc_f32(const float *v, int32_t *idx, int n) { stable_argsort_desc_f32(v, idx, n); }
void orc_stable_argsort_desc_f64(const double *v, int32_t *idx, int n) { stable_argsort_desc_f64(v, idx, n); }

/* ------------------------------------------------------------- TDM tree */

/* T/tree/DistTree.scala:26-38 (loadItems) + T/tree/TDMTree.scala:12-33: the
 * state a loaded TDMTree holds, as dense arrays instead of hash maps. */
typedef struct {
  int max_level;
  int32_t max_code;        /* codes.max over the id-code pairs   (DistTree.scala:36) */
  int32_t non_leaf_offset; /* leafIds.max + 1                    (DistTree.scala:35) */
  int64_t n_slots;         /* dense codeNodeMap domain */
  uint8_t *exists, *is_leaf;
  int32_t *node_id;
  int64_t n_ids;           /* dense idCodeMap domain = non_leaf_offset */
  int32_t *id_to_code;     /* -1 = not a key */
} orc_tree_t;

void *orc_tree_create(const int32_t *codes, const int32_t *ids, const uint8_t *is_leaf, int64_t n_nodes,
                      const int32_t *leaf_ids, const int32_t *leaf_codes, int64_t n_leaf, int max_level) {
  orc_tree_t *t = (orc_tree_t *)calloc(1, sizeof(*t));
  t->max_level = max_level;
  int32_t mc = -1;
  for (int64_t i = 0; i < n_nodes; i++) if (codes[i] > mc) mc = codes[i];
  for (int64_t i = 0; i < n_leaf; i++) if (leaf_codes[i] > mc) mc = leaf_codes[i];
  t->n_slots = (int64_t)mc + 1;
  t->exists = (uint8_t *)calloc(t->n_slots > 0 ? t->n_slots : 1, 1);
  t->is_leaf = (uint8_t *)calloc(t->n_slots > 0 ? t->n_slots : 1, 1);
  t->node_id = (int32_t *)calloc(t->n_slots > 0 ? t->n_slots : 1, sizeof(int32_t));
  for (int64_t i = 0; i < n_nodes; i++) {
    t->exists[codes[i]] = 1; t->is_leaf[codes[i]] = is_leaf[i] ? 1 : 0; t->node_id[codes[i]] = ids[i];
  }
  int32_t mid = -1, mlc = -1;
  for (int64_t i = 0; i < n_leaf; i++) { if (leaf_ids[i] > mid) mid = leaf_ids[i]; if (leaf_codes[i] > mlc) mlc = leaf_codes[i]; }
  t->non_leaf_offset = mid + 1;
  t->max_code = mlc;
  t->n_ids = t->non_leaf_offset;
  t->id_to_code = (int32_t *)malloc(sizeof(int32_t) * (t->n_ids > 0 ? t->n_ids : 1));
  for (int64_t i = 0; i < t->n_ids; i++) t->id_to_code[i] = -1;
  for (int64_t i = 0; i < n_leaf; i++) if (leaf_ids[i] >= 0) t->id_to_code[leaf_ids[i]] = leaf_codes[i];
  return t;
}
void orc_tree_destroy(void *p) {
  orc_tree_t *t = (orc_tree_t *)p;
  if (!t) return;
  free(t->exists); free(t->is_leaf); free(t->node_id); free(t->id_to_code); free(t);
}
int orc_tree_non_leaf_offset(void *p) { return ((orc_tree_t *)p)->non_leaf_offset; }
int orc_tree_max_code(void *p) { return ((orc_tree_t *)p)->max_code; }

static int tree_contains(const orc_tree_t *t, int64_t code) { return code >= 0 && code < t->n_slots && t->exists[code]; }

/* T/tree/TDMTree.scala:35-56 idToCode: returns the number of mask positions */
int orc_tdm_id_to_code(void *p, const int32_t *item_ids, int n, int32_t *codes, int32_t *mask_pos) {
  orc_tree_t *t = (orc_tree_t *)p;
  int nm = 0;
  for (int i = 0; i < n; i++) {
    int32_t id = item_ids[i];
    if (id == 0) { mask_pos[nm++] = i; codes[i] = -1; }                       /* paddingId -> paddingIdx */
    else if (id < t->non_leaf_offset && id >= 0 && t->id_to_code[id] >= 0) codes[i] = t->id_to_code[id];
    else {
      /* ancestors: wraps like JVM Int subtraction */
      int32_t tmp = (int32_t)((uint32_t)id - (uint32_t)t->non_leaf_offset);
      if (tmp > t->max_code) { mask_pos[nm++] = i; codes[i] = -1; }
      else codes[i] = tmp;
    }
  }
  return nm;
}

/* T/model/Recommender.scala:210-216 getLevelStart — the reference's own floating formula */
void orc_level_start(int candidate_num, int *start_code, int *level) {
  int lv = (int)floor(log((double)candidate_num) / log(2.0));
  int s = 0;
  for (int i = 1; i <= lv; i++) s = s * 2 + 1;
  *start_code = s; *level = lv;
}

/*
 * Scorer plug: scores `n` candidate node codes for ONE user whose (already
 * idToCode'd) history is seq_codes[L] with padding at mask_pos[n_mask].
 * Mirrors modelInputs.buildInputs + model.forward (Recommender.scala:93-94).
 */
typedef int (*orc_scorer_f32)(void *ctx, const int32_t *codes, int n, const int32_t *seq_codes, int L,
                              const int32_t *mask_pos, int n_mask, float *out);

/* Default scorer = the DIN restatement, fed exactly what duplicateSequence /
 * MaskModelInputs.buildInputs build (Recommender.scala:138-201): the history
 * replicated per candidate and the flat mask list m + i*L. */
int orc_din_scorer_f32(void *din, const int32_t *codes, int n, const int32_t *seq_codes, int L,
                       const int32_t *mask_pos, int n_mask, float *out) {
  int32_t *seqs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1) * L);
  int32_t *pads = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1) * (n_mask > 0 ? n_mask : 1));
  int64_t np = 0;
  for (int i = 0; i < n; i++) {
    memcpy(seqs + (size_t)i * L, seq_codes, sizeof(int32_t) * L);
    for (int m = 0; m < n_mask; m++) pads[np++] = mask_pos[m] + i * L;
  }
  int rc = orc_din_forward_f32(din, codes, seqs, pads, np, n, out);
  free(seqs); free(pads);
  return rc;
}

/*
 * One iteration of the level fold, integer logic only
 * (T/model/Recommender.scala:58-101, lines 62-92):
 *   in : candidates (code, pred) in order
 *   out: leaves of this level (appended, in candidate order, to leaf_codes/leaf_preds),
 *        children codes of the pruned beam, in beam order, filtered by existence.
 * Returns the number of children (0 => the fold's remaining iterations are no-ops).
 */
int orc_tdm_level_step(void *p, int beam, const int32_t *cand_codes, const float *cand_preds, int n_cand,
                       int32_t *leaf_codes, float *leaf_preds, int *n_leaf_out, int32_t *children) {
  orc_tree_t *t = (orc_tree_t *)p;
  int32_t *nl_codes = (int32_t *)malloc(sizeof(int32_t) * (n_cand > 0 ? n_cand : 1));
  float *nl_preds = (float *)malloc(sizeof(float) * (n_cand > 0 ? n_cand : 1));
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (n_cand > 0 ? n_cand : 1));
  int nl = 0, nleaf = 0;
  for (int i = 0; i < n_cand; i++) { /* partition(isLeaf), :62-64 */
    if (t->is_leaf[cand_codes[i]]) { leaf_codes[nleaf] = cand_codes[i]; leaf_preds[nleaf] = cand_preds[i]; nleaf++; }
    else { nl_codes[nl] = cand_codes[i]; nl_preds[nl] = cand_preds[i]; nl++; }
  }
  *n_leaf_out = nleaf;
  int nb = nl;
  if (nl > beam) { stable_argsort_desc_f32(nl_preds, order, nl); nb = beam; } /* :74-84 */
  else for (int i = 0; i < nl; i++) order[i] = i;                           /* :85-87 keeps input order */
  int nc = 0;
  for (int i = 0; i < nb; i++) { /* :88-92 */
    int64_t c = nl_codes[order[i]];
    if (tree_contains(t, 2 * c + 1)) children[nc++] = (int32_t)(2 * c + 1);
    if (tree_contains(t, 2 * c + 2)) children[nc++] = (int32_t)(2 * c + 2);
  }
  free(nl_codes); free(nl_preds); free(order);
  return nc;
}

/*
 * Final selection: T/model/Recommender.scala:103-106 (drop consumed ids, emit
 * (id, pred)) followed by T/model/TDM.scala:21 / Recommender.scala:36
 * (stable sortBy(-pred), take(topk)).  `leaf_*` must already be in the order of
 * the reference's leafNodes list (latest level first).
 */
int orc_tdm_finalize(void *p, const int32_t *leaf_codes, const float *leaf_preds, int n_leaf,
                     const int32_t *consumed, int n_consumed, int topk, int32_t *out_ids, float *out_preds) {
  orc_tree_t *t = (orc_tree_t *)p;
  int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (n_leaf > 0 ? n_leaf : 1));
  float *pr = (float *)malloc(sizeof(float) * (n_leaf > 0 ? n_leaf : 1));
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (n_leaf > 0 ? n_leaf : 1));
  int n = 0;
  for (int i = 0; i < n_leaf; i++) {
    int32_t id = t->node_id[leaf_codes[i]];
    int drop = 0;
    for (int c = 0; c < n_consumed; c++) if (consumed[c] == id) { drop = 1; break; }
    if (!drop) { ids[n] = id; pr[n] = leaf_preds[i]; n++; }
  }
  stable_argsort_desc_f32(pr, order, n);
  int k = n < topk ? n : topk;
  for (int i = 0; i < k; i++) { out_ids[i] = ids[order[i]]; out_preds[i] = pr[order[i]]; }
  free(ids); free(pr); free(order);
  return k;
}

/*
 * Recommender._recommend + TDM.recommend (T/model/Recommender.scala:40-107,
 * T/model/TDM.scala:17-22).  Output preds are LOGITS (the reference applies
 * sigmoid in double afterwards, TDM.scala:56-58; callers do that).
 *
 * trace (optional): for parity tests. trace_codes/trace_preds receive every
 * level's scored candidate list back to back, trace_counts[k] its length,
 * *trace_levels the number of scored levels.
 * Returns the number of recommendations (<= topk), negative on scorer error.
 */
int orc_tdm_recommend(void *p, orc_scorer_f32 scorer, void *ctx, const int32_t *seq_ids, int L, int topk,
                      int beam, int use_mask, const int32_t *consumed, int n_consumed, int32_t *out_ids,
                      float *out_preds, int32_t *trace_codes, float *trace_preds, int32_t *trace_counts,
                      int *trace_levels) {
  orc_tree_t *t = (orc_tree_t *)p;
  int32_t *seq_codes = (int32_t *)malloc(sizeof(int32_t) * L);
  int32_t *mask_pos = (int32_t *)malloc(sizeof(int32_t) * L);
  int n_mask = orc_tdm_id_to_code(t, seq_ids, L, seq_codes, mask_pos); /* duplicateSequence :171,181 */
  if (!use_mask) n_mask = 0;                                           /* SeqModelInputs passes no mask */
  int start, level;
  orc_level_start(beam, &start, &level);
  int cap = 2 * beam > 2 * (start + 1) ? 2 * beam : 2 * (start + 1);
  int n_iter = t->max_level - level + 1; /* (level to tree.maxLevel) */
  if (n_iter < 0) n_iter = 0;
  int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * cap);
  float *pred = (float *)malloc(sizeof(float) * cap);
  int32_t *child = (int32_t *)malloc(sizeof(int32_t) * cap);
  /* leafNodes list: level blocks are PREPENDED (leafNodes ++: levelInfo.leafNodes, :66-67) */
  int64_t leaf_cap = (int64_t)cap * (n_iter + 1);
  int32_t *lv_codes = (int32_t *)malloc(sizeof(int32_t) * leaf_cap);
  float *lv_preds = (float *)malloc(sizeof(float) * leaf_cap);
  int *lv_off = (int *)calloc(n_iter + 2, sizeof(int));
  int n_cand = 0, n_lv = 0, total_leaf = 0, tl = 0, toff = 0, rc = 0;
  for (int64_t c = start; c < 2 * (int64_t)start + 1; c++) /* :53-56 */
    if (tree_contains(t, c)) { cand[n_cand] = (int32_t)c; pred[n_cand] = 0.0f; n_cand++; }
  for (int it = 0; it < n_iter && n_cand > 0; it++) {
    int nleaf = 0;
    int nc = orc_tdm_level_step(t, beam, cand, pred, n_cand, lv_codes + total_leaf, lv_preds + total_leaf, &nleaf, child);
    lv_off[n_lv] = total_leaf; total_leaf += nleaf; n_lv++; lv_off[n_lv] = total_leaf;
    if (nc == 0) { n_cand = 0; break; }
    rc = scorer(ctx, child, nc, seq_codes, L, mask_pos, n_mask, pred);
    if (rc != 0) break;
    memcpy(cand, child, sizeof(int32_t) * nc);
    n_cand = nc;
    if (trace_counts) {
      memcpy(trace_codes + toff, cand, sizeof(int32_t) * nc);
      memcpy(trace_preds + toff, pred, sizeof(float) * nc);
      trace_counts[tl++] = nc; toff += nc;
    }
  }
  if (trace_levels) *trace_levels = tl;
  int k = 0;
  if (rc == 0) {
    /* NB: candidates still alive after the last iteration are dropped, exactly as the fold does */
    int32_t *fl_codes = (int32_t *)malloc(sizeof(int32_t) * (total_leaf > 0 ? total_leaf : 1));
    float *fl_preds = (float *)malloc(sizeof(float) * (total_leaf > 0 ? total_leaf : 1));
    int o = 0;
    for (int b = n_lv - 1; b >= 0; b--)
      for (int i = lv_off[b]; i < lv_off[b + 1]; i++) { fl_codes[o] = lv_codes[i]; fl_preds[o] = lv_preds[i]; o++; }
    k = orc_tdm_finalize(t, fl_codes, fl_preds, total_leaf, consumed, n_consumed, topk, out_ids, out_preds);
    free(fl_codes); free(fl_preds);
  }
  free(seq_codes); free(mask_pos); free(cand); free(pred); free(child); free(lv_codes); free(lv_preds); free(lv_off);
  return rc != 0 ? (rc < 0 ? rc : -rc) : k;
}

/* Recommender.recommendItems (T/model/Recommender.scala:18-37): the eval path
 * widens the beam for users with many consumed items. */
int orc_tdm_recommend_items(void *p, orc_scorer_f32 scorer, void *ctx, const int32_t *seq_ids, int L, int topk,
                            int beam, int use_mask, const int32_t *consumed, int n_consumed, int has_consumed,
                            int32_t *out_ids, float *out_preds) {
  int cn = beam;
  if (has_consumed) { int w = (n_consumed + topk) / 2; cn = w > beam ? w : beam; }
  else n_consumed = 0;
  return orc_tdm_recommend(p, scorer, ctx, seq_ids, L, topk, cn, use_mask, consumed, n_consumed, out_ids, out_preds,
                           NULL, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------ OTM */

typedef int (*orc_scorer_f64)(void *ctx, const int32_t *codes, int n, const int32_t *seq_codes, int L, double *out);

/* CandidateSearcher.buildInputs (O/model/CandidateSearcher.scala:85-107): history
 * replicated per candidate; mask = every flat position whose code is paddingIdx. */
int orc_din_scorer_f64(void *din, const int32_t *codes, int n, const int32_t *seq_codes, int L, double *out) {
  int32_t *seqs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1) * L);
  int32_t *pads = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1) * L);
  int64_t np = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < L; j++) {
      seqs[(size_t)i * L + j] = seq_codes[j];
      if (seq_codes[j] == -1) pads[np++] = i * L + j;
    }
  int rc = orc_din_forward_f64(din, codes, seqs, pads, np, n, out);
  free(seqs); free(pads);
  return rc;
}

/* O/package.scala:15-17 */
int orc_lower_log2(int n) { return (int)floor(log((double)n) / log(2.0)); }
int orc_upper_log2(int n) { return (int)ceil(log((double)n) / log(2.0)); }

/* CandidateSearcher.buildBeamNodes (O/model/CandidateSearcher.scala:109-122) */
int orc_otm_beam_nodes(const int32_t *ids, const double *scores, int n, int beam, int beam_start, int32_t *out) {
  int o = 0;
  if (beam_start) {
    for (int i = 0; i < n; i++) { out[o++] = ids[i] * 2 + 1; out[o++] = ids[i] * 2 + 2; }
  } else {
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    stable_argsort_desc_f64(scores, order, n);
    int nb = n < beam ? n : beam;
    for (int i = 0; i < nb; i++) { out[o++] = ids[order[i]] * 2 + 1; out[o++] = ids[order[i]] * 2 + 2; }
    free(order);
  }
  return o;
}

/*
 * CandidateSearcher.beamSearch (O/model/CandidateSearcher.scala:58-80): complete
 * tree, no existence filter, every level after the first sorts and keeps `beam`.
 * seq_codes are already node ids (OTM.recommend maps items through itemIdMapping,
 * O/model/OTM.scala:15).  Returns the number of leaf-level candidates.
 */
int orc_otm_beam_search(orc_scorer_f64 scorer, void *ctx, const int32_t *seq_codes, int L, int leaf_level, int beam,
                        int32_t *out_ids, double *out_scores) {
  int start_level = orc_lower_log2(beam);
  int start = 0;
  for (int i = 1; i <= start_level; i++) start = start * 2 + 1;
  int n = start + 1; /* Seq.range(startNode, startNode*2+1) */
  int cap = 2 * (n > beam ? n : beam);
  int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * cap);
  double *sc = (double *)malloc(sizeof(double) * cap);
  int32_t *child = (int32_t *)malloc(sizeof(int32_t) * cap);
  for (int i = 0; i < n; i++) { ids[i] = start + i; sc[i] = 0.0; }
  int rc = 0;
  for (int level = start_level; level < leaf_level; level++) {
    int nc = orc_otm_beam_nodes(ids, sc, n, beam, level == start_level, child);
    rc = scorer(ctx, child, nc, seq_codes, L, sc);
    if (rc != 0) break;
    memcpy(ids, child, sizeof(int32_t) * nc);
    n = nc;
  }
  if (rc == 0) { memcpy(out_ids, ids, sizeof(int32_t) * n); memcpy(out_scores, sc, sizeof(double) * n); }
  free(ids); free(sc); free(child);
  return rc != 0 ? (rc < 0 ? rc : -rc) : n;
}

/*
 * OTM.recommend tail (O/model/OTM.scala:17-21): keep candidates that map to an
 * item, stable sort by score descending, take topk.  node_to_item[node] = item
 * id or -1.  Scores returned are logits (sigmoid applied by the caller).
 */
int orc_otm_finalize(const int32_t *ids, const double *scores, int n, const int32_t *node_to_item, int64_t n_nodes,
                     int topk, int32_t *out_items, double *out_scores) {
  int32_t *it = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  double *sc = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; i++)
    if (ids[i] >= 0 && ids[i] < n_nodes && node_to_item[ids[i]] >= 0) { it[m] = node_to_item[ids[i]]; sc[m] = scores[i]; m++; }
  stable_argsort_desc_f64(sc, order, m);
  int k = m < topk ? m : topk;
  for (int i = 0; i < k; i++) { out_items[i] = it[order[i]]; out_scores[i] = sc[order[i]]; }
  free(it); free(sc); free(order);
  return k;
}

/* first n in [1, n_max] where the reference's floating getLevelStart level differs from
 * floor(log2 n) in integer arithmetic, or 0 when they agree everywhere */
int orc_level_start_first_mismatch(int n_max) {
  for (int n = 1; n <= n_max; n++) {
    int s, lv, il = 0;
    orc_level_start(n, &s, &lv);
    while ((2 << il) <= n) il++;
    if (il != lv || s != (1 << il) - 1) return n;
    if (orc_lower_log2(n) != il) return n;
  }
  return 0;
}

/* ------------------------------------------------------------ threaded batch (cpu_baseline) */
#include <malloc.h>
#include <pthread.h>

/* One worker per core over contiguous user ranges, as the reference's evaluator splits users over
 * its thread pool (T/evaluation/Evaluator.scala:28-37).  Each worker runs the plain per-user
 * orc_tdm_recommend with the DIN restatement as scorer. */
typedef struct {
  void *tree, *din;
  const int32_t *seqs;
  int64_t u0, u1;
  int L, topk, beam, use_mask;
  int32_t *out_ids; float *out_preds; int32_t *out_counts;
} orc_batch_job;

static void *orc_batch_worker(void *arg) {
  orc_batch_job *j = (orc_batch_job *)arg;
  for (int64_t u = j->u0; u < j->u1; u++) {
    int k = orc_tdm_recommend(j->tree, orc_din_scorer_f32, j->din, j->seqs + u * j->L, j->L, j->topk, j->beam,
                              j->use_mask, NULL, 0, j->out_ids + u * j->topk, j->out_preds + u * j->topk,
                              NULL, NULL, NULL, NULL);
    j->out_counts[u] = k;
  }
  return NULL;
}

int orc_tdm_recommend_batch(void *tree, void *din, const int32_t *seqs, int64_t U, int L, int topk, int beam,
                            int use_mask, int n_threads, int32_t *out_ids, float *out_preds, int32_t *out_counts) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > U) n_threads = (int)U;
  /* the per-call scratch buffers must come from per-thread arenas, not mmap/munmap (which serialise
   * every thread on the process address-space lock and dominate beyond ~32 threads) */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_ARENA_MAX, n_threads + 8);
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
  orc_batch_job *jobs = (orc_batch_job *)malloc(sizeof(orc_batch_job) * n_threads);
  for (int t = 0; t < n_threads; t++) {
    orc_batch_job j = {tree, din, seqs, U * t / n_threads, U * (t + 1) / n_threads, L, topk, beam, use_mask,
                       out_ids, out_preds, out_counts};
    jobs[t] = j;
    pthread_create(&th[t], NULL, orc_batch_worker, &jobs[t]);
  }
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return 0;
}

/* ------------------------------------------------------------------ JTM tree learning
 * J/ = jtm/src/main/scala/com/mass/jtm/
 * Restates J/optim/JTM.scala:22-73 (optimize), J/optim/TreeLearning.scala:48-194 (child weights) and
 * :217-265 (reBalance), J/tree/JTMTree.scala:36-113 (ancestor / idToCode).
 *
 * UNPINNED detail: JTM.optimize groups items through Scala immutable HashMaps
 * (`oldProjection.toArray.groupMap`, JTM.scala:31), so the order of `itemsAssignedToNode` — which only
 * breaks ties between items with EQUAL (moved?, weight) keys in reBalance — is the JVM's hash-trie
 * iteration order.  This restatement uses ascending item id (the order items are passed in).
 */

/* JTMTree.getAncestorAtLevel (J/tree/JTMTree.scala:36-43) on a node code */
static int32_t jtm_ancestor_code(int32_t code, int level) {
  const int64_t lim = ((int64_t)1 << (level + 1)) - 1;
  while (code >= lim) code = (code - 1) >> 1;
  return code;
}
int32_t orc_jtm_ancestor_at_level(void *p, int32_t item, int level) {
  orc_tree_t *t = (orc_tree_t *)p;
  return jtm_ancestor_code(t->id_to_code[item], level);
}

/* JTMTree.idToCodeWithMask (J/tree/JTMTree.scala:86-113): mask holds ONLY padding-id positions */
int orc_jtm_id_to_code_with_mask(void *p, const int32_t *ids, int n, int level, int hierarchical, int min_level,
                                 int32_t *codes, int32_t *mask_pos) {
  orc_tree_t *t = (orc_tree_t *)p;
  int nm = 0;
  for (int i = 0; i < n; i++) {
    int32_t id = ids[i];
    if (id == 0) { codes[i] = -1; mask_pos[nm++] = i; }
    else if (id < t->non_leaf_offset && id >= 0 && t->id_to_code[id] >= 0)
      codes[i] = (hierarchical && level >= min_level) ? jtm_ancestor_code(t->id_to_code[id], level) : t->id_to_code[id];
    else {
      int32_t c = (int32_t)((uint32_t)id - (uint32_t)t->non_leaf_offset);
      codes[i] = c > t->max_code ? -1 : c;
    }
  }
  return nm;
}

/* TreeLearning.aggregateWeights (J/optim/TreeLearning.scala:152-174) for one item and one child:
 * walk child -> up to (excl.) current node; each step = one forward over the item's rows and a
 * sequential float sum of the logits (Tensor.sum).  row_ids: [n_rows * L] raw item ids. */
static float jtm_aggregate(orc_tree_t *t, void *din, const int32_t *row_ids, int n_rows, int L, int32_t current_node,
                           int32_t child, int level, int hierarchical, int min_level, int use_mask) {
  if (n_rows == 0) return -1e6f;
  float weights = 0.0f;
  int32_t node = child;
  int lv = level;
  int32_t *codes = (int32_t *)malloc(sizeof(int32_t) * n_rows);
  int32_t *seqs = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows * L);
  int32_t *mask = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows * L);
  float *out = (float *)malloc(sizeof(float) * n_rows);
  while (node > current_node) {
    for (int i = 0; i < n_rows; i++) codes[i] = node;
    int nm = orc_jtm_id_to_code_with_mask(t, row_ids, n_rows * L, lv, hierarchical, min_level, seqs, mask);
    if (!use_mask) nm = 0;
    orc_din_forward_f32(din, codes, seqs, mask, nm, n_rows, out);
    float score = 0.0f;
    for (int i = 0; i < n_rows; i++) score += out[i];   /* DenseTensorMath.scala:395-406, sequential */
    weights += score;
    node = (node - 1) / 2;
    lv -= 1;
  }
  free(codes); free(seqs); free(mask); free(out);
  return weights;
}

/* child weights of every item of ONE gap step.  items[n]: item ids; row_off[n+1] / row_ids: the item's
 * training rows (itemSequenceMap, TreeLearning.scala:34-46); item_node[n]: node (at old_level) the item
 * currently sits in.  weights[n * nchild], child order = JTMTree.getChildrenAtLevel (J/tree/JTMTree.scala:53-57). */
int orc_jtm_child_weights(void *tree, void *din, const int32_t *items, const int64_t *row_off, const int32_t *row_ids,
                          const int32_t *item_node, int64_t n, int L, int old_level, int level, int hierarchical,
                          int min_level, int use_mask, float *weights) {
  orc_tree_t *t = (orc_tree_t *)tree;
  const int nchild = 1 << (level - old_level);
  for (int64_t i = 0; i < n; i++) {
    const int64_t first = ((int64_t)item_node[i] << (level - old_level)) + nchild - 1;   /* leftmost descendant */
    for (int c = 0; c < nchild; c++)
      weights[i * nchild + c] = jtm_aggregate(t, din, row_ids + row_off[i] * L, (int)(row_off[i + 1] - row_off[i]), L,
                                              item_node[i], (int32_t)(first + c), level, hierarchical, min_level, use_mask);
  }
  return 0;
}

/*
 * getChildrenProjection minus the scoring (J/optim/TreeLearning.scala:48-97) for ONE parent node:
 * sortNodeWeights (stable, descending, :137-150), first choice = best child (:66-71), reBalance (:217-265).
 * items[n] (in the order the reference would iterate them), weights[n * nchild], old_node[n] =
 * tree.getAncestorAtLevel(item, level).  out_node[n] = assigned child code, or -1 when the greedy loop
 * drops the item (every remaining candidate already processed).
 */
int orc_jtm_rebalance(const int32_t *items, const float *weights, const int32_t *old_node, int64_t n, int32_t node,
                      int old_level, int level, int max_assign, int32_t *out_node) {
  const int nchild = 1 << (level - old_level);
  const int64_t first = ((int64_t)node << (level - old_level)) + nchild - 1;
  /* candidateNodeWeights(item): children sorted by weight desc, stable */
  int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * nchild);
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * nchild);
  for (int64_t i = 0; i < n; i++) {
    stable_argsort_desc_f32(weights + i * nchild, order, nchild);
    for (int c = 0; c < nchild; c++) cand[i * nchild + c] = order[c];
  }
  /* per child: member list of (item index, weight, nextWeightIdx) */
  typedef struct { int64_t it; float w; int next; } info_t;
  info_t **lst = (info_t **)calloc(nchild, sizeof(info_t *));
  int64_t *cnt = (int64_t *)calloc(nchild, sizeof(int64_t)), *capv = (int64_t *)calloc(nchild, sizeof(int64_t));
  uint8_t *present = (uint8_t *)calloc(nchild, 1), *processed = (uint8_t *)calloc(nchild, 1);
#define PUSH(c, IT, W, NX) do { if (cnt[c] == capv[c]) { capv[c] = capv[c] ? capv[c] * 2 : 16; lst[c] = (info_t *)realloc(lst[c], sizeof(info_t) * capv[c]); } \
                                 lst[c][cnt[c]].it = (IT); lst[c][cnt[c]].w = (W); lst[c][cnt[c]].next = (NX); cnt[c]++; present[c] = 1; } while (0)
  for (int64_t i = 0; i < n; i++) { int c = cand[i * nchild]; PUSH(c, i, weights[i * nchild + c], 1); }
  for (;;) {
    /* getMaxNode (:203-215): first child in order with the largest member count among unprocessed, present ones */
    int64_t best = -1; int bc = 0;
    for (int c = 0; c < nchild; c++) {
      int64_t v = (!processed[c] && present[c]) ? cnt[c] : -1;
      if (c == 0 || v > best) { best = v; bc = c; }
    }
    if (best <= max_assign) break;
    processed[bc] = 1;
    /* sortBy (moved?, weight desc), stable (:240-242) */
    const int64_t m = cnt[bc];
    info_t *src = lst[bc];
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * m), *tmp = (int64_t *)malloc(sizeof(int64_t) * m);
    for (int64_t i = 0; i < m; i++) idx[i] = i;
    const int32_t this_code = (int32_t)(first + bc);
    for (int64_t w = 1; w < m; w *= 2) {
      for (int64_t lo = 0; lo < m; lo += 2 * w) {
        int64_t mid = lo + w < m ? lo + w : m, hi = lo + 2 * w < m ? lo + 2 * w : m, a = lo, b = mid, o = lo;
        while (a < mid && b < hi) {
          const info_t *x = &src[idx[a]], *y = &src[idx[b]];
          int mx = old_node[x->it] != this_code, my = old_node[y->it] != this_code;
          int cmp = mx != my ? (mx < my ? -1 : 1) : java_float_compare(y->w, x->w);   /* false < true; weight reversed */
          if (cmp > 0) tmp[o++] = idx[b++]; else tmp[o++] = idx[a++];
        }
        while (a < mid) tmp[o++] = idx[a++];
        while (b < hi) tmp[o++] = idx[b++];
      }
      memcpy(idx, tmp, sizeof(int64_t) * m);
    }
    info_t *sorted = (info_t *)malloc(sizeof(info_t) * m);
    for (int64_t i = 0; i < m; i++) sorted[i] = src[idx[i]];
    free(idx); free(tmp);
    cnt[bc] = max_assign;
    memcpy(lst[bc], sorted, sizeof(info_t) * max_assign);
    for (int64_t i = max_assign; i < m; i++) {   /* redundant items, in order */
      const info_t it = sorted[i];
      for (int k = it.next; k < nchild; k++) {
        int c = cand[it.it * nchild + k];
        if (!processed[c]) { PUSH(c, it.it, weights[it.it * nchild + c], k + 1); break; }
      }
    }
    free(sorted);
  }
#undef PUSH
  for (int64_t i = 0; i < n; i++) out_node[i] = -1;
  for (int c = 0; c < nchild; c++)
    for (int64_t k = 0; k < cnt[c]; k++) out_node[lst[c][k].it] = (int32_t)(first + c);
  for (int c = 0; c < nchild; c++) free(lst[c]);
  free(lst); free(cnt); free(capv); free(present); free(processed); free(cand); free(order);
  (void)items;
  return 0;
}


/* ------------------------------------------------------------- level-wise negative sampling (row A10)
 * NegativeSampler.sample / sampleFromUniformDistribution / sampleFromCategoricalDistribution
 * (T/utils/NegativeSampler.scala:76-158) + MiniBatch.convert / transformWithMask (T/dataset/MiniBatch.scala:49-88,129-147).
 * The reference draws from an unseeded ThreadLocalRandom / MersenneTwister(System.nanoTime()), so no draw-for-draw parity
 * with the JVM exists; this restatement keeps the reference's ALGORITHM (per level: distinct existing codes != the positive,
 * emitted after the positive in ascending order; categorical mode: at most negNum + tolerance draws from the level's
 * node-probability distribution, then a uniform fill that — as in the reference, :131-137 — does not exclude the positive)
 * and plugs in the product's counter-based stream splitmix64(seed, target, level, draw), so that the device sampler can be
 * compared with it bit for bit.
 */
static uint64_t orc_splitmix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint64_t orc_draw(uint64_t seed, int64_t t, int level, uint64_t ctr, int stream) {
  return orc_splitmix(seed ^ orc_splitmix((uint64_t)t * 64 + (uint64_t)level + ((uint64_t)stream << 40)) ^ (ctr * 0xD6E8FEB86659FD93ull));
}
static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return x < y ? -1 : x > y; }

/* node_codes/node_probs [n_prob]: Node.probality of every tree node (only read when with_prob).  Outputs as
 * dm_tdm_make_train_batch: out_codes/out_labels/out_rowmask [rows], out_seqs [rows*L]; returns the number of rows. */
int64_t orc_tdm_sample_batch(void *tree, const int32_t *seq_item_ids, const int32_t *target_item_ids, int64_t T, int L,
                             const int32_t *neg_counts, int start_level, uint64_t seed, int use_mask, int with_prob,
                             int tolerance, const int32_t *node_codes, const float *node_probs, int64_t n_prob,
                             int32_t *out_codes, int32_t *out_seqs, uint32_t *out_rowmask, float *out_labels) {
  orc_tree_t *t = (orc_tree_t *)tree;
  const int total_level = t->max_level + 1;
  /* levelProbs (:59-66): existing codes of a level in ascending order with their probabilities; EnumeratedIntegerDistribution
   * normalises them and samples by inverse CDF */
  float *prob_of = NULL;
  if (with_prob) {
    prob_of = (float *)calloc(t->n_slots > 0 ? t->n_slots : 1, sizeof(float));
    for (int64_t i = 0; i < n_prob; i++) if (node_codes[i] >= 0 && node_codes[i] < t->n_slots) prob_of[node_codes[i]] = node_probs[i];
  }
  int32_t *seq_codes = (int32_t *)malloc(sizeof(int32_t) * L);
  int32_t *negs = (int32_t *)malloc(sizeof(int32_t) * 4096);
  int64_t q = 0;
  for (int64_t ti = 0; ti < T; ti++) {
    uint32_t m = 0;
    for (int j = 0; j < L; j++) {     /* TDMTree.idToCode, T/tree/TDMTree.scala:35-56 */
      int32_t id = seq_item_ids[ti * L + j], cd;
      if (id == 0) { cd = -1; m |= 1u << j; }
      else if (id < t->non_leaf_offset && id >= 0 && t->id_to_code[id] >= 0) cd = t->id_to_code[id];
      else { cd = (int32_t)((uint32_t)id - (uint32_t)t->non_leaf_offset); if (cd > t->max_code) { cd = -1; m |= 1u << j; } }
      seq_codes[j] = cd;
    }
    int32_t tid = target_item_ids[ti], tcode = -1;
    if (tid > 0 && tid < t->non_leaf_offset && t->id_to_code[tid] >= 0) tcode = t->id_to_code[tid];
    if (tcode <= 0 || !tree_contains(t, tcode)) continue;      /* pathNodes of an unknown target is empty (:70-77) */
    int depth = 0;
    for (int64_t c = tcode; c > 0; c = (c - 1) >> 1) depth++;
    for (int level = start_level; level < total_level && level <= depth; level++) {
      int64_t pos = tcode;
      for (int d = depth; d > level; d--) pos = (pos - 1) >> 1;
      const int64_t lo = ((int64_t)1 << level) - 1, width = (int64_t)1 << level;
      const int want = neg_counts[level];
      int n = 0;
      if (with_prob) {
        /* the level's distribution */
        int64_t ne = 0;
        for (int64_t c = lo; c < lo + width; c++) if (tree_contains(t, c)) ne++;
        int32_t *codes = (int32_t *)malloc(sizeof(int32_t) * (ne > 0 ? ne : 1));
        double *cdf = (double *)malloc(sizeof(double) * (ne > 0 ? ne : 1));
        double sum = 0.0;
        int64_t k = 0;
        for (int64_t c = lo; c < lo + width; c++) if (tree_contains(t, c)) { codes[k] = (int32_t)c; sum += (double)prob_of[c]; k++; }
        double acc = 0.0;
        for (k = 0; k < ne; k++) { acc += (double)prob_of[codes[k]] / sum; cdf[k] = acc; }
        for (uint64_t ctr = 0; n < want && ctr < (uint64_t)(want + tolerance); ctr++) {       /* :120-126 */
          const double u = (double)(orc_draw(seed, ti, level, ctr, 0) >> 11) * (1.0 / 9007199254740992.0);
          int64_t a = 0, b = ne;                    /* first index with u < cdf[index]; the last one when there is none */
          while (a < b) { int64_t mid = (a + b) >> 1; if (u < cdf[mid]) b = mid; else a = mid + 1; }
          if (a >= ne) a = ne - 1;
          const int32_t s = codes[a];
          int dup = 0;
          for (int i = 0; i < n; i++) dup |= negs[i] == s;
          if (!dup && s != pos) negs[n++] = s;
        }
        free(codes); free(cdf);
        /* :127-139 — the uniform fill after the tolerance is used up does NOT exclude the positive (reference behaviour) */
        const uint64_t limit = 64ull * (uint64_t)(want + 1) + 4096ull * (uint64_t)width;
        for (uint64_t ctr = 0; n < want && ctr <= limit; ctr++) {
          const int64_t sc = lo + (int64_t)(orc_draw(seed, ti, level, ctr, 1) % (uint64_t)width);
          if (!tree_contains(t, sc)) continue;
          int dup = 0;
          for (int i = 0; i < n; i++) dup |= negs[i] == (int32_t)sc;
          if (!dup) negs[n++] = (int32_t)sc;
        }
      } else {
        /* :146-158; the product bounds the rejection loop (a level with fewer than want + 1 existing nodes would spin forever) */
        const uint64_t limit = 64ull * (uint64_t)(want + 1) + 4096ull * (uint64_t)width;
        for (uint64_t ctr = 0; n < want && ctr <= limit; ctr++) {
          const int64_t sc = lo + (int64_t)(orc_draw(seed, ti, level, ctr, 0) % (uint64_t)width);
          if (sc == pos || !tree_contains(t, sc)) continue;
          int dup = 0;
          for (int i = 0; i < n; i++) dup |= negs[i] == (int32_t)sc;
          if (!dup) negs[n++] = (int32_t)sc;
        }
      }
      qsort(negs, n, sizeof(int32_t), cmp_i32);          /* BitSet.toList: ascending */
      for (int k = 0; k <= n; k++, q++) {
        out_codes[q] = k == 0 ? (int32_t)pos : negs[k - 1];
        out_labels[q] = k == 0 ? 1.0f : 0.0f;
        memcpy(out_seqs + q * L, seq_codes, sizeof(int32_t) * L);
        out_rowmask[q] = use_mask ? m : 0u;
      }
    }
  }
  free(seq_codes); free(negs); free(prob_of);
  return q;
}

/* ------------------------------------------------------------- Deep-Retrieval (row A13) */
#include "dr_body.inc"
