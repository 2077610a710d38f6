/*
 * dismember_hip.h — C ABI of libdismember_hip.so, the MI355X (gfx950) native
 * implementation of dismember's tree beam-search retrieval hot path.
 *
 * This is the drop-in boundary: every entry point names the
 * reference interface it replaces.  The reference has no FFI of its own for
 * this path (it is pure Scala calling MKL through BigDL's JNI); these are the
 * functions a JNI shim binds so that the Scala facades
 *   TDM.recommend            tdm/src/main/scala/com/mass/tdm/model/TDM.scala:17-22
 *   Recommender.recommendItems  tdm/.../model/Recommender.scala:18-37
 *   OTM.recommend            otm/src/main/scala/com/mass/otm/model/OTM.scala:14-22
 *   Module.forward           scalann/.../nn/abstractnn/AbstractModule.scala:19-41
 * keep their signatures.
 *
 * Conventions
 *   - every function returns 0 (DM_OK) or a negative dm_status; the message of
 *     the last failure on a handle is dm_last_error(h).  No exception crosses
 *     the boundary.
 *   - plain pointers and sizes only.  Host pointers are caller-owned and only
 *     read/written during the call.  Device memory is owned by the handle.
 *   - one handle per (device, stream); a handle is thread-compatible (one host
 *     thread at a time), like one cloned Module in the reference
 *     (tdm/.../optim/LocalOptimizer.scala:28-44).
 *   - there is NO CPU fallback: without a usable HIP device dm_create fails.
 *   - T/ = tdm/src/main/scala/com/mass/tdm/, O/ = otm/src/main/scala/com/mass/otm/,
 *     S/ = scalann/src/main/scala/com/mass/scalann/ in the citations below.
 */
#ifndef DISMEMBER_HIP_H
#define DISMEMBER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dm_ctx *dm_handle_t;

typedef enum {
  DM_OK = 0,
  DM_ERR_INVALID = -1,     /* bad argument (the reference's `require` failures) */
  DM_ERR_HIP = -2,         /* HIP runtime error */
  DM_ERR_STATE = -3,       /* tree / weights not loaded yet */
  DM_ERR_INDEX = -4,       /* embedding index out of range (LookupTable.scala:47-53 throws) */
  DM_ERR_UNSUPPORTED = -5  /* shape outside what the kernels are built for */
} dm_status;

typedef enum { DM_F32 = 0, DM_F64 = 1 } dm_dtype;

/* ---- lifetime ---------------------------------------------------------- */
int dm_version(void);
int dm_device_count(int *count);
int dm_create(int device_id, dm_handle_t *out);
int dm_destroy(dm_handle_t h);
/* A second handle on the same device that READS the tree, the id maps, the weights and every derived copy (fragment orders, split
 * planes, the pre-split table) of `h` — the same device memory, nothing is copied — through its own stream, request arenas and
 * workspace: the reference's `cloneModule()` workers share one weight storage (tdm/.../optim/LocalOptimizer.scala:28-44, pinned by
 * otm/src/test/scala/CloneModelSpec.scala:20-36).  One host thread per handle, as everywhere; searches through `h` and its clones run
 * concurrently.  Loading and training go through the owning handle only (DM_ERR_STATE on a clone); a clone sees the new weights at its
 * next call.  Weight updates must not overlap a clone's search in flight (the reference's workers meet at a barrier before the update,
 * LocalOptimizer.scala:73-80).  Destroy the clones before the owner (dm_destroy(owner) fails with DM_ERR_STATE while clones exist).
 * A clone of a clone shares the same owner.  The Deep-Retrieval model is per handle (not shared). */
int dm_clone(dm_handle_t h, dm_handle_t *clone);

const char *dm_last_error(dm_handle_t h); /* h may be NULL: last create-time error */
int dm_synchronize(dm_handle_t h);        /* hipStreamSynchronize on the handle's stream */

/* ---- index structures (replaces TDMOp.tree, T/operator/TDMOp.scala:18-19,61-82) ---- */

/* codeNodeMap (T/tree/DistTree.scala:40-87): one entry per tree node.
 * codes[i] = heap code, node_ids[i] = Node.id, is_leaf[i] = Node.is_leaf. */
int dm_load_tree_tdm(dm_handle_t h, const int32_t *codes, const int32_t *node_ids, const uint8_t *is_leaf,
                     int64_t n_nodes, int max_level);
/* idCodeMap + nonLeafOffset + maxCode (DistTree.loadItems, T/tree/DistTree.scala:26-38) */
int dm_load_id_maps(dm_handle_t h, const int32_t *leaf_item_ids, const int32_t *leaf_codes, int64_t n);
/* TDM.loadTree / TDMOp.initTree(treePbPath) (T/model/TDM.scala:50-52, T/operator/TDMOp.scala:61-82): the reference's own tree file —
 * [int32 BE length][KVItem] records written by TreeBuilder.build (T/tree/TreeBuilder.scala:23-101) / JTMTree.writeTree, read by
 * DistTree.loadData (T/tree/DistTree.scala:40-87) — parsed in the library: dm_load_tree_tdm + dm_load_id_maps + dm_tdm_set_node_probs. */
int dm_load_tree_file(dm_handle_t h, const char *path);
/* TDMTree.idToCode (T/tree/TDMTree.scala:35-56); host-side helper, same logic the kernels run on device */
int dm_tdm_id_to_code(dm_handle_t h, const int32_t *item_ids, int n, int32_t *codes, int32_t *mask_pos,
                      int *n_mask);
/* Recommender.getLevelStart (T/model/Recommender.scala:210-216) in integer arithmetic */
int dm_level_start(int candidate_num, int *start_code, int *level);

/* ---- scorer weights (replaces Serialization.loadModel + DIN.buildModel) ---- */

/* compact parameter vector in Graph.parameters order (S/nn/graphnn/Graph.scala:37-48,
 * S/nn/mixin/Module.scala:9-45; DIN: T/model/DIN.scala:18-42):
 *   [emb num_index x E ; att.W E x E ; l1.W E x 2E ; l1.b E ; l2.W 1 x E ; l2.b 1]
 * E: any size 1..128 (the reference's embedSize is free).  The kernels are built for 16 / 32 / 64 / 128; other sizes are zero-padded to
 * the next of them on the device (same function: zero columns change no product, the softmax scale stays 1 / sqrt(E)); every vector
 * that crosses the boundary (this one, dm_train_download, the checkpoint) keeps the MODEL's layout.  dm_load_weights_din_dev* and the
 * raw device-pointer exchange (dm_train_dense_block / _export_rows / _add_rows) take native sizes only. */
int dm_load_weights_din(dm_handle_t h, int dtype, int E, int64_t num_index, const void *compact,
                        int64_t n_elems);

/* Checkpoint (replaces TDM.saveModel / loadModel, T/utils/Serialization.scala:60-101 — a Java ObjectOutputStream of the module graph
 * there; tdm/src/test/scala/TdmModelTrainSpec.scala:85-96 pins save -> load -> identical recommendations): one flat little-endian
 * file with the compact parameter vector in the loaded dtype plus the tree nodes and the item id -> leaf code map when they are
 * loaded (layout: dismember_amd/csrc/checkpoint.hip.inc).  No optimizer state, like the reference.  dm_load_model replaces whatever
 * tree / id maps / weights the handle holds with the file's. */
int dm_save_model(dm_handle_t h, const char *path);
int dm_load_model(dm_handle_t h, const char *path);

/* Arithmetic of the beam-search scorer (dm_tdm_beam_search*, dm_otm_beam_search*; no reference counterpart: the
 * reference's Linear / MatMul call MKL sgemm, S/tensor/TensorNumeric.scala:265-266).
 *   DM_SCORER_F32        fp32-input MFMA for every product.
 *   DM_SCORER_SPLIT_F16  the three products of a scored row (attention scores q.k, the W1a half and the attention half of
 *                        linear1) take each fp32 operand as hi + lo fp16 halves (22 significand bits, scaled by a power of
 *                        two into the fp16 range) and run hi*hi + hi*lo + lo*hi on the fp16 matrix pipe with fp32
 *                        accumulation: per-product relative error <= 3 * 2^-22, below the fp32 summation-order noise that
 *                        separates any two fp32 implementations over E terms (measured: DESIGN.md).  E must be 32, 64 or
 *                        128.  Same stated tolerance as the F32 mode (rtol 1e-4 / atol 1e-5 against the oracle); ids are
 *                        the exact beam search on the scores produced.
 *   DM_SCORER_AUTO       (default) SPLIT_F16 where the embedding size allows it, F32 otherwise (E = 16); and for the OTM
 *                        searches (dm_otm_beam_search*) of a model loaded as DM_F64: DM_SCORER_F64.
 *   DM_SCORER_F64        OTM searches in the reference's own arithmetic (otm/.../model/DIN.scala:12-39 is DIN[Double]):
 *                        every product on the fp64 matrix cores, node ids bit-exact against the fp64 oracle given the
 *                        same scores, scores within 1e-10 / 1e-9.  Needs f64 weights.  This is the parity mode; F32 /
 *                        SPLIT_F16 (on the f32 copy of the table) remain the throughput modes and can be forced.
 * The brute-force recall oracle (dm_tdm_bruteforce_topk) and every other entry point always use fp32 / fp64 arithmetic.
 * dm_get_scorer_mode: the setting, the arithmetic in effect for the loaded model, and the power-of-two shifts in use
 * (after the first search). */
enum { DM_SCORER_F32 = 0, DM_SCORER_SPLIT_F16 = 1, DM_SCORER_AUTO = 2, DM_SCORER_F64 = 3 };
int dm_set_scorer_mode(dm_handle_t h, int mode);
int dm_get_scorer_mode(dm_handle_t h, int *mode, int *effective, int *shift_emb, int *shift_w);

/* ---- operator level: Module.forward(Table(items, seqs, masks)) ---------- */

/* call sites: T/model/Recommender.scala:93-94, O/model/CandidateSearcher.scala:40-50,
 * O/tree/OTMTree.scala:167-172, jtm/.../optim/TreeLearning.scala:166-168.
 * codes [B], seqs [B*L] (node codes, -1 = padding), pad_flat_idx [n_pad] = flat
 * positions i*L+j to mask (S/nn/Mask.scala:27-32).  logits: B values of the
 * loaded dtype (float or double). */
int dm_din_forward(dm_handle_t h, const int32_t *codes, const int32_t *seqs, const int32_t *pad_flat_idx,
                   int64_t n_pad, int64_t B, int L, void *logits);

/* ---- TDM beam search: Recommender._recommend + TDM.recommend -------------- */

typedef struct {
  int beam;           /* candidateNum */
  int topk;
  int use_mask;       /* 1 for DIN (TDM.apply, T/model/TDM.scala:26-29) */
  int widen_consumed; /* 1 = recommendItems semantics: beam_u = max((consumed_u+topk)/2, beam)  (:28-33) */
} dm_tdm_search_opts;

/* seq_item_ids [U*L] raw item ids (0 = padding).  consumed_off [U+1] / consumed_ids: CSR of
 * already-consumed item ids per user, or NULL/NULL.  Outputs: out_item_ids/out_scores [U*topk]
 * (scores are LOGITS; the facade applies sigmoid in double like T/model/TDM.scala:56-58),
 * out_counts [U] (<= topk).
 * L = seq_len, 1 .. 32 (the reference's Attention takes any length, S/nn/Attention.scala:34-53; its configs use 10): up to 16 positions run on
 * the one-wave kernel, 17 .. 32 on the fused LDS-fed kernel's two-key-tile instance (f32 models; fp64 OTM searches on the fp64 kernel's);
 * L > 32 is DM_ERR_INVALID, never a truncated history. */
int dm_tdm_beam_search(dm_handle_t h, const int32_t *seq_item_ids, int64_t U, int L, const dm_tdm_search_opts *opts,
                       const int64_t *consumed_off, const int32_t *consumed_ids, int32_t *out_item_ids,
                       float *out_scores, int32_t *out_counts);

/* Same search, additionally dumping every scored level (parity instrumentation):
 * trace_codes/trace_scores [U * max_levels * cap], trace_counts [U * max_levels];
 * cap = 2*beam rounded up to 16; slots past a user's last level have count 0. */
int dm_tdm_beam_search_trace(dm_handle_t h, const int32_t *seq_item_ids, int64_t U, int L,
                             const dm_tdm_search_opts *opts, int32_t *out_item_ids, float *out_scores,
                             int32_t *out_counts, int max_levels, int32_t *trace_codes, float *trace_scores,
                             int32_t *trace_counts);

/* ---- OTM beam search: CandidateSearcher.beamSearch (O/model/CandidateSearcher.scala:58-80) ----
 * seq_codes [U*L] are node ids (OTM.recommend has already mapped items, O/model/OTM.scala:15);
 * complete tree, no existence filter.  out_node_ids/out_scores [U * 2*beam] = the leaf-level
 * candidates in order (f32 scorer; the reference runs this path in f64 — tolerance in DESIGN.md). */
int dm_otm_beam_search(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                       int32_t *out_node_ids, float *out_scores, int32_t *out_counts);

/* The same two searches with double outputs, always in fp64 arithmetic (f64 weights required): the scores the reference's
 * DIN[Double] produces, for the 1e-10 / 1e-9 parity contract. */
int dm_otm_beam_search_f64(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                           int32_t *out_node_ids, double *out_scores, int32_t *out_counts);
int dm_otm_beam_search_trace_f64(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                                 int32_t *out_node_ids, double *out_scores, int32_t *out_counts, int max_levels,
                                 int32_t *trace_codes, double *trace_scores, int32_t *trace_counts);

/* Same, dumping every level's scored candidates = OTMTree.beamSearchNodes (O/tree/OTMTree.scala:67-91), which OTM
 * training consumes: trace_* [U * max_levels * cap] / [U * max_levels], cap = 2*beam rounded up to 16 (min 32). */
int dm_otm_beam_search_trace(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                             int32_t *out_node_ids, float *out_scores, int32_t *out_counts, int max_levels,
                             int32_t *trace_codes, float *trace_scores, int32_t *trace_counts);

/* ---- brute force over every leaf (build-defined recall@k oracle) ---- */
int dm_tdm_bruteforce_topk(dm_handle_t h, const int32_t *seq_item_ids, int64_t U, int L, int topk, int use_mask,
                           int32_t *out_item_ids, float *out_scores, int32_t *out_counts);

/* ---- JTM / OTM tree learning (jtm/src/main/scala/com/mass/jtm/optim/TreeLearning.scala) ---------------
 * One gap step of JTM.optimize (jtm/.../optim/JTM.scala:29-70) for n_items items:
 *   row_off [n_items+1] / row_item_ids [rows*L]: the item's training rows = itemSequenceMap (:34-46), raw item
 *   ids, L per row; item_node [n_items]: node code (at old_level) each item currently sits in.
 * weights [n_items * 2^(level-old_level)] = aggregateWeights (:152-174) per child, children ordered as
 * JTMTree.getChildrenAtLevel (jtm/.../tree/JTMTree.scala:53-57); items without rows get -1e6 (:160).
 * The DIN forwards run on the GPU; Tensor.sum and the chain accumulation keep the reference's fp32 order. */
int dm_jtm_child_weights(dm_handle_t h, const int64_t *row_off, const int32_t *row_item_ids, const int32_t *item_node,
                         int64_t n_items, int L, int old_level, int level, int hierarchical, int min_level, int use_mask,
                         float *weights);
/* The same with the catalogue's training rows (itemSequenceMap, built once per run: TreeLearning.scala:34-46) kept on the device
 * across the gap steps of one JTM.optimize: dm_jtm_cache_rows uploads row_off [n_items+1] (row_off[0] == 0) / row_item_ids once
 * (n_items == 0 drops the copy); dm_jtm_child_weights_cached scores the items [i_lo, i_lo + n_items) of that catalogue
 * (item_node and weights are indexed from i_lo: a rank's item shard) and uploads nothing but item_node. */
int dm_jtm_cache_rows(dm_handle_t h, const int64_t *row_off, const int32_t *row_item_ids, int64_t n_items, int L);
/* Sharded runs (one handle per rank, dm_jtm_optimize_cached with a communicator attached / dm_jtm_optimize_all): a rank only ever scores
 * the items [i_lo, i_hi) that dm_jtm_shard_range(n_items, rank, nranks, ...) gives it (the reference's contiguous split of the items
 * over its workers, JTM.scala:47-52).  dm_jtm_cache_rows_range takes the SAME full arrays and keeps row_off for all items, but uploads
 * only that range's training rows — the upload, the row arrays and the per-row history codes then scale with 1 / nranks.  Scoring items
 * outside the range on such a handle is DM_ERR_STATE. */
int dm_jtm_shard_range(int64_t n_items, int rank, int nranks, int64_t *i_lo, int64_t *i_hi);
int dm_jtm_cache_rows_range(dm_handle_t h, const int64_t *row_off, const int32_t *row_item_ids, int64_t n_items, int L, int64_t i_lo, int64_t i_hi);
int dm_jtm_child_weights_cached(dm_handle_t h, const int32_t *item_node, int64_t i_lo, int64_t n_items, int old_level, int level,
                                int hierarchical, int min_level, int use_mask, float *weights);
/* One whole gap step of JTM.optimize (JTM.scala:36-72) over the cached catalogue on the device: the child weights of every item
 * (as dm_jtm_child_weights_cached) and the greedy re-balance of every parent node of old_level (as dm_jtm_rebalance_all) with the
 * [n_items x 2^gap] weight matrix kept in HBM.  item_node [n_items] = current node of every item, old_node [n_items] =
 * tree.getAncestorAtLevel(item, level); out_node [n_items] = the item's node at `level` after the step.  n_items must be the
 * cached catalogue's size (single-rank runs; sharded runs keep the two separate calls around their all-gather). */
int dm_jtm_step_cached(dm_handle_t h, const int32_t *item_node, const int32_t *old_node, int64_t n_items, int old_level, int level,
                       int hierarchical, int min_level, int use_mask, int max_assign, int32_t *out_node);
/* JTM.optimize (JTM.scala:22-73) over the cached catalogue in ONE call: every item starts at the root; each gap step scores the children
 * chains and re-balances every parent node on the device, and its result feeds the next step without leaving HBM.  item_code [n_items] =
 * the items' current leaf codes (tree.getAncestorAtLevel is taken from them per step); out_proj [n_items] = every item's new leaf code;
 * step_seconds (may be NULL) [2] = seconds spent scoring / re-balancing.
 * With a communicator attached (dm_comm_attach, > 1 rank) the call is COLLECTIVE and the run is sharded the way the reference's workers
 * split it (JTM.scala:33-68, JTMAsync.scala:43-75): every rank holds the cached catalogue and scores the rows of ITS contiguous item range
 * (sizes as taskSize / extraSize, :47-52); the [items x 2^gap] weight slices are all-gathered in place, device to device (RCCL over xGMI;
 * host staging on DM_COMM_HOST); the greedy re-balance runs replicated while a level has fewer parent nodes than ranks and afterwards
 * over each rank's contiguous range of parent nodes, followed by an all-gather of (item, new node) pairs.  Each weight is one GPU's
 * sequential fp32 sum and each parent's list keeps item order, so out_proj is bit-identical to the single-rank result on every rank. */
int dm_jtm_optimize_cached(dm_handle_t h, const int32_t *item_code, int64_t n_items, int max_level, int gap, int hierarchical, int min_level,
                           int use_mask, int32_t *out_proj, double *step_seconds);
/* the same from ONE process driving all ranks of a dm_comm_create_all clique (the reference's shape: one JVM, numThreads workers):
 * hs[i] carries rank i and its own cached catalogue; ranks run on host threads of this call.  The ranks' projections are compared
 * (DM_ERR_STATE if they differ) and out_proj receives the common one.  n == 1 is dm_jtm_optimize_cached. */
int dm_jtm_optimize_all(dm_handle_t *hs, int n, const int32_t *item_code, int64_t n_items, int max_level, int gap, int hierarchical,
                        int min_level, int use_mask, int32_t *out_proj, double *step_seconds);
/* measurement — what the last dm_jtm_optimize_cached on this handle did: out10[0..9] = ranks, transport (DM_COMM_*; 2^64-1 = no
 * communicator), items this rank scored summed over the gap steps, items it re-balanced in node-sharded steps, steps with a replicated
 * re-balance, node-sharded steps, weight bytes all-gathered, projection bytes all-gathered, 0, 0; secs3 = scoring, re-balance,
 * exchange seconds. */
int dm_jtm_optimize_stats(dm_handle_t h, uint64_t *out10, double *secs3);
/* measurement: seconds the last dm_jtm_step_cached spent in its scoring pass and in its re-balance (copies included) */
int dm_jtm_last_step_seconds(dm_handle_t h, double *scoring_s, double *rebalance_s);
/* getChildrenProjection after scoring (:58-97) for the items of ONE parent `node`: sortNodeWeights (stable
 * descending), first choice, greedy capacity-bounded reBalance (:217-265).  old_node [n] =
 * tree.getAncestorAtLevel(item, level); out_node [n] = assigned child code (-1: dropped by the greedy loop).
 * Host-side exact integer logic (parents are independent; callers may run it from several threads on
 * different handles). */
int dm_jtm_rebalance(dm_handle_t h, const float *weights, const int32_t *old_node, int64_t n, int32_t node, int old_level,
                     int level, int max_assign, int32_t *out_node);

/* The same for EVERY parent node of a level in one call (the loop over currentNodes of JTM.optimize, J/optim/JTM.scala:36-70;
 * TreeConstruction.run, O/tree/TreeConstruction.scala:44-101): items are grouped by item_node [n] (the node each item sits in at
 * old_level) in the order given; out_node [n] = the new child code, or the old node for items the greedy loop drops. */
int dm_jtm_rebalance_all(dm_handle_t h, const float *weights, const int32_t *old_node, const int32_t *item_node, int64_t n,
                         int old_level, int level, int max_assign, int32_t *out_node);
int dm_otm_rebalance_all(dm_handle_t h, const double *weights, const int32_t *old_node, const int32_t *item_node, int64_t n,
                         int old_level, int level, int max_assign, int32_t *out_node);

/* OTM twin (otm/src/main/scala/com/mass/otm/tree/TreeConstruction.scala): the item sequences hold NODE ids (-1 =
 * paddingIdx, :218-232), the scorer runs in the loaded dtype (fp64 in the reference) and sums are double:
 * dm_otm_child_weights = aggregateWeights (:194-212) per child of getChildrenAtLevel (:281-285);
 * dm_otm_rebalance     = sortNodeWeights (:180-192) + first choice + reBalance (:304-352), same logic as the JTM one. */
int dm_otm_child_weights(dm_handle_t h, const int64_t *row_off, const int32_t *row_codes, const int32_t *item_node,
                         int64_t n_items, int L, int old_level, int level, int use_mask, double *weights);
int dm_otm_rebalance(dm_handle_t h, const double *weights, const int32_t *old_node, int64_t n, int32_t node, int old_level,
                     int level, int max_assign, int32_t *out_node);

/* ---- training step (tdm/src/main/scala/com/mass/tdm/optim/LocalOptimizer.scala:58-187) ------------------
 * One handle == one worker (the reference's per-thread model clone).  A step is
 *   dm_train_forward_backward  == trainBatch (:139-162): zero-initialised gradients accumulate the mean-BCE
 *                                  gradient of this worker's rows (BCECriterionWithLogits + Module.backward)
 *   [exchange between workers]  == syncGradients (:164-187): sum over workers, then / n_workers
 *   dm_adam_step(1/n_workers)   == Adam.optimize (scalann/.../optim/Adam.scala:19-73), DENSE over the whole
 *                                  compact vector (untouched embedding rows still move through s, r), then
 *                                  the gradient is zeroed (zeroGradParameters).
 * The step runs in the LOADED dtype: f32 models (TDM / JTM, T/model/DIN.scala) train in fp32, f64 models (the reference's OTM,
 * DIN[Double]: O/model/DIN.scala:12-39, O/optim/LocalOptimizer.scala:55-140,217-233) in fp64 on v_mfma_f64_16x16x4_f64 with an
 * fp64 gradient and fp64 Adam state.  Labels are float in both (0 / 1, or pseudo targets clipped to [0, 1]); `loss` is the
 * value rounded to float, dm_train_last_loss returns it in the model's precision.  Vectors that cross the boundary
 * (dm_train_download, dm_train_dense_block, dm_train_export_rows, dm_train_add_rows) hold elements of the loaded dtype. */
typedef struct { double lr, lr_decay, beta1, beta2, eps; } dm_adam_opts;   /* reference defaults: 1e-3, 0, 0.9, 0.999, 1e-8 */
int dm_train_init(dm_handle_t h, const dm_adam_opts *opts);
int dm_train_forward_backward(dm_handle_t h, const int32_t *codes, const int32_t *seqs, const int32_t *pad_flat_idx,
                              int64_t n_pad, const float *labels, int64_t B, int L, float *loss);
int dm_adam_step(dm_handle_t h, float grad_scale);
/* The step visits the embedding rows a gradient has ever reached since dm_train_init plus the small matrices: for every other row
 * g = s = r = 0 and the reference's dense update leaves the weight bit-identical (0 / (sqrt(0) + eps) = 0, w + (-0) = w), so the result IS
 * the dense Adam's, without streaming a table that cannot change (dense stream again once a quarter of the rows are active, when eps == 0,
 * or with DM_ADAM_DENSE=1 in the environment).  Measurement: rows visited by the last step, and whether it took the active-rows path. */
int dm_adam_last_step_rows(dm_handle_t h, uint64_t *rows, int *active_rows_only);
int dm_train_last_loss(dm_handle_t h, double *loss);
/* what: 0 weights, 1 gradient, 2 Adam s, 3 Adam r — host copy of the full compact-layout vector (tests, checkpoints) */
int dm_train_download(dm_handle_t h, int what, void *out, int64_t n);
/* Gradient exchange for N workers on N GPUs: the dense block [att.W ; l1.W ; l1.b ; l2.W ; l2.b]
 * is all-reduced in place through the returned device pointer (RCCL); embedding gradients are row-sparse:
 * export this worker's unique touched rows (index + gradient row, device buffers), all-gather them, add the
 * other workers' rows.  Replicas stay bit-identical because every rank applies the same sums in the same order. */
int dm_train_dense_block(dm_handle_t h, void **d_ptr, int64_t *n);
int dm_train_export_rows(dm_handle_t h, int32_t *d_rows, void *d_grads, int64_t cap, int64_t *n);   /* NULL buffers: size query */
int dm_train_add_rows(dm_handle_t h, const int32_t *d_rows, const void *d_grads, int64_t n);

/* ---- multi-GPU: communicators and the gradient exchange ------------------------------------------
 * The reference's data parallelism is N worker threads in one JVM whose gradient buffers are averaged in place
 * (LocalOptimizer.syncGradients, T/optim/LocalOptimizer.scala:164-187; O/optim/LocalOptimizer.scala:217-233).  Here a worker
 * is a handle on its own GPU; a dm_comm_t connects the workers:
 *   DM_COMM_RCCL  RCCL over xGMI, one rank per GPU.  Either one process per GPU (dm_comm_create_rccl with an id from
 *                 dm_comm_unique_id that the host distributes itself, or dm_comm_create_tcp which distributes it), or ONE
 *                 process driving every GPU like the reference's JVM (dm_comm_create_all + dm_allreduce_grads).
 *   DM_COMM_HOST  a TCP star through rank 0, device buffers staged through host memory: several workers sharing one GPU
 *                 (RCCL refuses two ranks per device) and CPU-only processes for the host-buffer collectives.
 * dm_train_sync_gradients(h) == syncGradients minus the division (dm_adam_step(1/N) folds it in): all-reduce(sum) of the
 * dense block + all-gather of (touched row, gradient row) lists, each touched row rebuilt as 0 + g_0 + g_1 + ... in rank
 * order, so every replica ends with bit-identical gradients (the RCCL dense sum is identical on all ranks but in RCCL's
 * reduction order, not rank order; the HOST transport sums in rank order).  RCCL transport: one ncclAllGather of 16-byte
 * {count, ok} records + ONE host read-back, then one ncclAllReduce and one ncclAllGather of max-padded row blocks.
 * Collective: every rank must call it; a rank whose touched-row list overflowed makes EVERY rank return DM_ERR_STATE.
 * dm_train_sync_stats: what the last exchange on this handle moved — out[0..7] = nranks, transport, touched rows of this
 * rank, touched rows of all ranks, bytes sent, bytes received, host synchronisations, 0. */
typedef struct dm_comm *dm_comm_t;
enum { DM_COMM_HOST = 0, DM_COMM_RCCL = 1 };
#define DM_COMM_ID_BYTES 128
int dm_comm_unique_id(void *id128);                                  /* ncclGetUniqueId, padded to DM_COMM_ID_BYTES */
int dm_comm_create_rccl(int nranks, int rank, const void *id128, int device_id, dm_comm_t *out);
/* rendezvous at addr:port (rank 0 listens); transport DM_COMM_RCCL: the sockets only carry the id.  device_id is ignored
 * by DM_COMM_HOST (no GPU needed for the host-buffer collectives). */
int dm_comm_create_tcp(int nranks, int rank, const char *addr, int port, int transport, int device_id, dm_comm_t *out);
int dm_comm_create_all(int n, const int *devices, dm_comm_t *out /* [n] */);   /* ncclCommInitAll: one process, n GPUs */
int dm_comm_destroy(dm_comm_t c);
const char *dm_comm_last_error(dm_comm_t c);                          /* c may be NULL: last create-time error */
int dm_comm_rank(dm_comm_t c, int *rank, int *nranks, int *transport);
int dm_comm_barrier(dm_comm_t c);
int dm_comm_allreduce_f64(dm_comm_t c, double *vals, int n, int op /* 0 sum (rank order), 1 max */);   /* host values, in place */
/* var-size all-gather of host buffers: recv gets the blocks in rank order, sizes[nranks] their byte lengths; recv == NULL
 * only fills sizes */
int dm_comm_all_gather_v(dm_comm_t c, const void *send, size_t bytes, void *recv, size_t recv_cap, uint64_t *sizes);
int dm_comm_attach(dm_handle_t h, dm_comm_t c);                       /* the handle does not own c; NULL detaches */
int dm_comm_all_gather_dev(dm_handle_t h, const void *d_send, size_t bytes, void *d_recv, size_t recv_cap, uint64_t *sizes);
int dm_train_sync_gradients(dm_handle_t h);
int dm_train_sync_stats(dm_handle_t h, uint64_t *out8);
/* SURVEY.md §8b `dm_allreduce_grads(h[], n_gpu)`: the same exchange for ALL ranks of a dm_comm_create_all clique from one
 * thread (hs[i] must carry rank i); n == 1 is dm_train_sync_gradients. */
int dm_allreduce_grads(dm_handle_t *hs, int n);

/* ---- OTM training iteration (otm/src/main/scala/com/mass/otm/optim/LocalOptimizer.scala:55-109) ---------------------------
 * One LocalOptimizer iteration for this worker's batch of U users, every list on the device:
 *   targets   OTMTree.optimalPseudoTargets (O/tree/OTMTree.scala:27-46; computeTargets :104-129 with its mirrored prediction offsets,
 *             computeChildrenScores :131-165) — target_mode 0 — or OTMTree.normalTargets (:50-63) — target_mode 1;
 *   beams     OTMTree.beamSearchNodes (:67-91), weights fixed (the per-level trace of the OTM beam search);
 *   per level MiniBatch.batchTransform (O/dataset/MiniBatch.scala:17-40) -> trainBatch (LocalOptimizer.scala:111-131) ->
 *             syncGradients (:133-141: dm_train_sync_gradients when a communicator with > 1 rank is attached — the call is then
 *             COLLECTIVE) -> Adam.optimize with 1 / ranks.
 * seq_codes [U*L] node ids (-1 = padding); target_off [U+1] / target_nodes: CSR of every user's target LEAF nodes (OTMSample.targetItems
 * after the item -> node mapping); level_losses [leaf_level - floor(log2 beam)] receives the per-level losses averaged over the ranks
 * (levelLoss of :73-80, first level first), *n_levels their number.  Needs dm_train_init; runs in the loaded dtype (the reference's
 * OTM is DIN[Double]: fp64 search, fp64 training); the Adam step of every level moves the weights, like the reference. */
typedef struct {
  int beam;          /* beamSize */
  int leaf_level;    /* leaf level of the complete tree; the start level is floor(log2 beam) */
  int use_mask;      /* 1 for DIN */
  int target_mode;   /* 0 "pseudo" (optimalPseudoTargets), 1 "normal" (normalTargets) */
} dm_otm_train_opts;
int dm_otm_train_batch(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, const int64_t *target_off, const int32_t *target_nodes,
                       const dm_otm_train_opts *opts, double *level_losses, int *n_levels);
/* the target lists alone (parity instrumentation; no training state needed): out_nodes / out_labels [levels x target_off[U]] in the CSR of
 * target_off (a user's list at a level occupies the first out_counts[level][u] of its slots; unused slots hold -1 / 0), out_counts
 * [levels x U]; level index li <-> tree level floor(log2 beam) + 1 + li. */
int dm_otm_pseudo_targets(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, const int64_t *target_off, const int32_t *target_nodes,
                          const dm_otm_train_opts *opts, int32_t *out_nodes, double *out_labels, int32_t *out_counts);
/* measurement — the last dm_otm_train_batch on this handle: out6 = users, levels, rows of the pseudo-target forwards, rows trained, 0, 0;
 * secs5 = pseudo targets, beam search, forward / backward, gradient exchange, Adam (seconds) */
int dm_otm_train_stats(dm_handle_t h, uint64_t *out6, double *secs5);

/* Level-wise negative sampling + batch expansion, ON THE DEVICE: NegativeSampler.sample
 * (tdm/.../utils/NegativeSampler.scala:76-158: uniform and sample_with_probability modes) + MiniBatch.convert
 * (tdm/.../dataset/MiniBatch.scala:49-88).  One wave per (target, level); counter-based stream splitmix64(seed, target,
 * level, draw) — the reference's RNG is unseeded, so parity with it is distributional; the CPU oracle reproduces this
 * stream bit for bit.
 * seq_item_ids [T*L], target_item_ids [T]; neg_counts = model.layer_negative_counts (>= max_level+1 entries).
 * Rows per target = sum_{l=start_level}^{level of the target} (1 + neg_counts[l]) (targets outside the tree: none);
 * out_codes == NULL returns the upper bound T * sum_{l=start_level}^{max_level} (1 + neg_counts[l]) in *n_rows.
 * out_rowmask bit j = history position j masked; seq_len <= 32. */
typedef struct {
  int start_level;     /* model.start_sample_level (>= 1) */
  int with_prob;       /* model.sample_with_probability: draw from the level's node probabilities (dm_tdm_set_node_probs) */
  int tolerance;       /* model.sample_tolerance: categorical draws per level = neg + tolerance, then a uniform fill */
  int use_mask;        /* 1 for DIN */
  uint64_t seed;
} dm_sample_opts;
/* Node.probality of the tree nodes (T/tree/DistTree.scala:60-75; levelProbs, NegativeSampler.scala:59-66) */
int dm_tdm_set_node_probs(dm_handle_t h, const int32_t *codes, const float *probs, int64_t n);
/* host buffers in and out (the sampling itself runs on the device) */
int dm_tdm_make_train_batch(dm_handle_t h, const int32_t *seq_item_ids, const int32_t *target_item_ids, int64_t T, int L,
                            const int32_t *neg_counts, int n_counts, const dm_sample_opts *opts, int32_t *out_codes,
                            int32_t *out_seqs, uint32_t *out_rowmask, float *out_labels, int64_t cap, int64_t *n_rows);
/* device buffers in and out: the rows land where dm_train_forward_backward_dev reads them; neg_counts is a host array */
int dm_tdm_sample_train_batch_dev(dm_handle_t h, const int32_t *d_seq_item_ids, const int32_t *d_target_item_ids, int64_t T, int L,
                                  const int32_t *neg_counts, int n_counts, const dm_sample_opts *opts, int32_t *d_codes,
                                  int32_t *d_seqs, uint32_t *d_rowmask, float *d_labels, int64_t cap, int64_t *n_rows);
/* dm_train_forward_backward on rows that already live in device memory (the output of dm_tdm_sample_train_batch_dev or of
 * a caller's own sampler).  Asynchronous unless `loss` is non-NULL.  The ids are NOT range-checked (the host-buffer entry
 * point validates them like LookupTable.scala:29-53): every code / history entry must be -1 or in [0, num_index). */
int dm_train_forward_backward_dev(dm_handle_t h, const int32_t *d_codes, const int32_t *d_seqs, const uint32_t *d_rowmask,
                                  const float *d_labels, int64_t B, int L, float *loss);
/* The same step for a batch in which every user contributes `rows_per_user` candidate rows that share the user's history — what
 * MiniBatch.batchTransform (otm/.../dataset/MiniBatch.scala:17-40) and MiniBatch.transformWithMask (tdm/.../dataset/MiniBatch.scala:
 * 129-147) build by replicating the history per row: d_seq_codes [U][L] (-1 = padding), d_user_mask [U] (bit j: position j masked;
 * NULL = nothing masked), d_codes / d_labels [U * rows_per_user] user-major (code -1 = a zero item row).  Same loss and gradients as
 * dm_train_forward_backward_dev on the expanded rows (fp64 sums in a different order); an f64 model with L <= 16 takes the per-user
 * kernels (train_grouped_f64.hip.inc), anything else is expanded to plain rows.  Ids are not range-checked. */
int dm_train_forward_backward_grouped_dev(dm_handle_t h, const int32_t *d_seq_codes, const uint32_t *d_user_mask, const int32_t *d_codes,
                                          const float *d_labels, int64_t U, int rows_per_user, int L, float *loss);
int dm_memcpy_d2d(dm_handle_t h, void *dst, const void *src, size_t bytes);

/* ---- device-resident variants (bench: inputs already in HBM when the clock starts) ---- */
int dm_dev_alloc(dm_handle_t h, size_t bytes, void **dptr);
int dm_dev_free(dm_handle_t h, void *dptr);
int dm_memcpy_h2d(dm_handle_t h, void *dst, const void *src, size_t bytes);
int dm_memcpy_d2h(dm_handle_t h, void *dst, const void *src, size_t bytes);
/* asynchronous on the handle's stream; all pointers are device pointers; consumed_* may be NULL */
int dm_tdm_beam_search_dev(dm_handle_t h, const int32_t *d_seq_item_ids, int64_t U, int L,
                           const dm_tdm_search_opts *opts, const int64_t *d_consumed_off,
                           const int32_t *d_consumed_ids, int32_t *d_out_item_ids, float *d_out_scores,
                           int32_t *d_out_counts);
/* OTM.recommend's search (dm_otm_beam_search) on a device-resident request: d_seq_codes [U][L] node ids (-1 = padding; codes
 * outside the table count as padding), outputs [U][2*beam] leaf-level node ids (-1 filled) and scores, [U] counts. */
int dm_otm_beam_search_dev(dm_handle_t h, const int32_t *d_seq_codes, int64_t U, int L, int beam, int leaf_level,
                           int32_t *d_out_node_ids, float *d_out_scores, int32_t *d_out_counts);

/* ---- Deep-Retrieval serving ----
 * D/ = deep-retrieval/src/main/scala/com/mass/dr/.  Item ids here are the INTERNAL ids of
 * MappingOp.itemIdMapping (D/model/DeepRetrieval.scala:32,45 map in and out; the host facade keeps doing that);
 * -1 is paddingIdx (D/package.scala:19). */
typedef struct dm_dr_model {
  int32_t dtype;        /* DM_F32 | DM_F64: element type of every array below AND the arithmetic type of the path
                         * (the reference computes in fp64: TensorNumeric.NumericDouble, D/model/LayerModel.scala:7) */
  int32_t on_device;    /* 0: host arrays; 1: device arrays (copied device-to-device, the caller keeps ownership) */
  int32_t embed;        /* embedSize, multiple of 16 */
  int32_t seq_len;      /* seqLen */
  int32_t num_node;     /* K = numNode */
  int32_t num_layer;    /* D = numLayer, 2..4 (D/model/LayerModel.scala:12 requires >= 2) */
  int64_t num_item;
  const void *layer_emb;        /* LayerModel.embedParams   [(num_item + K(D-1)) x E]      D/model/LayerModel.scala:15,24 */
  const void *const *layer_w;   /* [D] LayerModel.linearParams(d).head  [K x (seq_len+d)E] D/model/LayerModel.scala:16-18,31-36 */
  const void *const *layer_b;   /* [D] LayerModel.linearParams(d).last  [K] */
  const void *rerank_emb;       /* RerankModel.embedParams  [num_item x E]                 D/model/RerankModel.scala:13 */
  const void *rerank_w;         /* RerankModel.linearParams.head [E x seq_len*E]           D/model/RerankModel.scala:14,32-34 */
  const void *rerank_b;         /* RerankModel.linearParams.last [E] */
  const void *softmax_w;        /* RerankModel.softmaxWeights [num_item x E]               D/model/RerankModel.scala:15 */
  const void *softmax_b;        /* RerankModel.softmaxBiases  [num_item]                   D/model/RerankModel.scala:16 */
                                /* the five rerank arrays may all be NULL: beam search only */
} dm_dr_model;
/* replaces the state DeepRetrieval.loadModel / LayerModel / RerankModel hold (D/model/DeepRetrieval.scala:90-106);
 * also precomputes the per-(layer, position) node tables, see DESIGN.md.  Float models with E % 64 == 0 additionally keep a second
 * copy of the item / node embedding rows (4 E bytes per row, as the first) split into fp16 hi/lo records for the 256 x 256 history GEMM
 * of batches that fill whole rounds of its tiles (4 096 users, 8 192 and more at config 5's shape); DM_DR_GEMM_X=0 in the environment at load time skips that copy and its kernel. */
int dm_dr_load_model(dm_handle_t h, const dm_dr_model *model);
/* MappingOp.pathItemMapping (D/model/MappingOp.scala:14-28) as a CSR over DISTINCT paths: path_nodes [n_paths x D],
 * items of path i = items[item_off[i] .. item_off[i+1]) in the order searchCandidate should yield them
 * (D/model/CandidateSearcher.scala:14-19).  Any path order; duplicate paths -> DM_ERR_INVALID. */
int dm_dr_load_path_items(dm_handle_t h, const int32_t *path_nodes, int64_t n_paths, const int64_t *item_off,
                          const int32_t *items);
/* CandidateSearcher.beamSearch (D/model/CandidateSearcher.scala:22-60) for U users: seq_ids [U x seq_len] internal ids;
 * out_paths [U x beam x D] nodes in layer order (-1 beyond the count), out_probs [U x beam] path probabilities in
 * descending order (ties: earlier parent path, then lower node — the stable sortBy of :49-51), out_counts [U]. */
int dm_dr_beam_search(dm_handle_t h, const int32_t *seq_ids, int64_t U, int beam, int32_t *out_paths, double *out_probs,
                      int32_t *out_counts);
/* DeepRetrieval.recommend (D/model/DeepRetrieval.scala:26-46) on internal ids: searchCandidate + RerankModel.inference
 * (D/model/RerankModel.scala:43-52) + stable descending sort + take(topk).  out_ids [U x topk] internal item ids
 * (-1 beyond the count; an item reachable through several top paths appears once per path, as in the reference),
 * out_scores [U x topk] rerank LOGITS (the reference applies sigmoid in double afterwards, :45). */
int dm_dr_recommend(dm_handle_t h, const int32_t *seq_ids, int64_t U, int beam, int topk, int32_t *out_ids,
                    double *out_scores, int32_t *out_counts);
/* device-resident variants (asynchronous on the handle's stream; ids are NOT range-checked) */
int dm_dr_beam_search_dev(dm_handle_t h, const int32_t *d_seq_ids, int64_t U, int beam, int32_t *d_out_paths,
                          double *d_out_probs, int32_t *d_out_counts);
int dm_dr_recommend_dev(dm_handle_t h, const int32_t *d_seq_ids, int64_t U, int beam, int topk, int32_t *d_out_ids,
                        double *d_out_scores, int32_t *d_out_counts);

/* ---- synthetic-data helpers (bench / tests only; nothing in the reference corresponds) ---- */
/* fill d_ptr[0..n) (float, device) with N(mean, std): counter-based splitmix64 + Box-Muller, reproducible per (seed, index) */
int dm_fill_normal(dm_handle_t h, float *d_ptr, int64_t n, float mean, float std, uint64_t seed);
/* the same values (the f32 draws, widened) into a double buffer: an fp64 model with the weights of the f32 one */
int dm_fill_normal_f64(dm_handle_t h, double *d_ptr, int64_t n, float mean, float std, uint64_t seed);
/* tree-correlated table for a heap of `depth` levels below the root: row(root) ~ N(0, std),
 * row(c) = rho * row(parent(c)) + sqrt(1 - rho^2) * N(0, std) — a stand-in for a trained index, where
 * a node's embedding summarises its subtree (with iid rows beam search has nothing to follow and
 * recall vs brute force is ~0 by construction). */
int dm_fill_tree_normal(dm_handle_t h, float *d_emb, int E, int depth, float rho, float std, uint64_t seed);
/* like dm_load_weights_din(DM_F32) but the compact vector already lives in device memory; the handle
 * takes ownership of d_compact (freed with the handle / on the next load). */
int dm_load_weights_din_dev(dm_handle_t h, int E, int64_t num_index, float *d_compact, int64_t n_elems);
/* the same for an fp64 model (dm_load_weights_din(DM_F64): the reference's OTM scorer is DIN[Double]) */
int dm_load_weights_din_dev_f64(dm_handle_t h, int E, int64_t num_index, double *d_compact, int64_t n_elems);

/* ---- measurement ----------------------------------------------------------- */
/* HIP events on the handle's stream around every beam-search kernel launched since the last
 * reset: number of launches and their summed duration. */
/* HIP-event pairs around the kernels launched since the last reset (at most the 4096 most recent launches) */
int dm_kernel_timing_reset(dm_handle_t h);
int dm_kernel_timing_get(dm_handle_t h, int *launches, double *total_ms);
/* the same, one kind of launch only: 0 = the search kernels proper, 1 = the second pass over the users the
 * one-wave-per-SIMD beam kernel hands to the LDS-fed kernel (one launch per search, empty on most inputs).
 * A Deep-Retrieval search is several launches: by default ONE pair brackets the whole search (kind 0); with
 * DM_DR_TIME_LAUNCHES=1 in the environment (read per call) every launch gets its own pair — kind 0 the history
 * GEMM, 11 layer 0, 10 + 2d / 11 + 2d the statistics / cut of layer d — at 2 - 14 % of the search's wall time.
 * Kind 30: the general-rows scorer (dm_din_forward, JTM child weights: dm_din_rows_split*_kernel), one pair per launch. */
int dm_kernel_timing_get_kind(dm_handle_t h, int kind, int *launches, double *total_ms);
/* name (with template arguments) of the kernel that ran the last TDM / OTM beam search, as a profiler lists it
 * (owned by the handle, valid until the next search) */
const char *dm_last_beam_kernel(dm_handle_t h);
/* scored rows (node, user) pairs of the last beam-search call, for roofline accounting */
int dm_last_scored_rows(dm_handle_t h, int64_t *rows);

#ifdef __cplusplus
}
#endif
#endif /* DISMEMBER_HIP_H */
