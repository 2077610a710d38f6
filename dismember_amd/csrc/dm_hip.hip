// libdismember_hip.so — MI355X (gfx950) implementation of dismember's tree beam-search
// retrieval hot path behind the C ABI of include/dismember_hip.h.
//
// No CPU fallback lives here: every entry point that computes launches HIP kernels.
#include "../../include/dismember_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <mutex>
#include <thread>
#include <utility>
#include <cerrno>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// Development builds (tools/build_probe.sh DM_DEV_LIGHT ...): only the E = 128 instances of the kernel templates are compiled — a third of
// the compile time while iterating on one kernel.  Never the product build: models of other embedding sizes fail with "unsupported".
#ifdef DM_DEV_LIGHT
#define DM_IF_ALL_E(...)
#else
#define DM_IF_ALL_E(...) __VA_ARGS__
#endif

#include "beam_kernel.hip.inc"
#include "beam_kernel_w.hip.inc"
#include "beam_kernel_f64.hip.inc"
#include "rows_kernel.hip.inc"
#include "train_kernel.hip.inc"
#include "train_grouped_f64.hip.inc"
#include "dr_kernel.hip.inc"
#include "dr_sliced.hip.inc"

#define DM_VERSION 100

// ------------------------------------------------------------------ context
struct dm_dr_state;
static void dm_dr_free(dm_dr_state *s);

// what the last gradient exchange on a handle moved (comm.hip.inc; dm_train_sync_stats)
struct dm_sync_stats { uint64_t rows_mine, rows_total, bytes_sent, bytes_recv; int host_syncs, nranks, transport; };

// what the last dm_jtm_optimize_cached on a handle did (jtm_sharded.hip.inc; dm_jtm_optimize_stats)
struct dm_jtm_stats {
  int nranks = 1, transport = -1, steps_replicated = 0, steps_node_sharded = 0;
  uint64_t items_scored = 0, items_rebalanced = 0, weight_bytes = 0, proj_bytes = 0;
  double scoring_s = 0, rebalance_s = 0, exchange_s = 0;
};

// what the last dm_otm_train_batch on a handle did (otm_train.hip.inc; dm_otm_train_stats)
struct dm_otm_stats {
  uint64_t users = 0, target_rows = 0, train_rows = 0;
  int levels = 0;
  double targets_s = 0, beam_s = 0, fwdbwd_s = 0, exchange_s = 0, adam_s = 0;
};

struct dm_ctx {
  int device = 0;
  int n_cu = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // dm_clone: a clone reads the tree, the weights and every derived copy of its parent (same device memory) through its own stream,
  // request arenas and workspace.  model_epoch counts the changes of anything a clone mirrors; mu serialises the lazy rebuilds.
  dm_ctx *parent = nullptr;
  std::atomic<int> n_clones{0};
  std::atomic<uint64_t> model_epoch{1};
  uint64_t seen_epoch = 0;
  uint64_t ids_epoch = 1;      // generation of the id maps (dm_load_id_maps) and the tree bitmaps (dm_load_tree_tdm)
  uint64_t seen_ids_epoch = 0; // clone: the owner's generation its host copies were taken at
  std::recursive_mutex mu;
  // tree (codeNodeMap as bitmaps + dense node-id array)
  bool tree_loaded = false, ids_loaded = false, leaves_at_max_only = true;
  int max_level = 0;
  int64_t n_slots = 0, n_leaf_nodes = 0;
  uint32_t *d_exists = nullptr, *d_leaf = nullptr;
  int32_t *d_node_id = nullptr, *d_id_to_code = nullptr, *d_leaf_codes = nullptr;
  int32_t non_leaf_offset = -1, max_code = -1;
  std::vector<int32_t> h_id_to_code;
  std::vector<uint32_t> h_exists;      // host copy of the existence bitmap (negative sampling)
  // weights
  bool w_loaded = false;
  int dtype = DM_F32, embed = 0;
  int embed_log = 0;           // the MODEL's embed size; embed is the kernels' (16 / 32 / 64 / 128): other sizes are zero-padded at load
  int64_t num_index = 0;
  void *d_compact = nullptr;   // as loaded (float or double)
  float *d_emb32 = nullptr;    // f32 table (aliases d_compact for DM_F32)
  bool emb32_owned = false;
  f32x4 *d_wfrag = nullptr;
  f32x4 *d_afrag = nullptr, *d_bfrag = nullptr;
  f32x4 *d_attA = nullptr, *d_w1aA = nullptr, *d_w1bA = nullptr;   // A-fragment order (rows kernel)
  float *d_b1 = nullptr, *d_w2 = nullptr;
  float b2 = 0.f;
  // split-fp16 scorer (dm_set_scorer_mode): fp16 hi/lo planes of W1a and the power-of-two scales, rebuilt lazily
  int scorer_mode = DM_SCORER_AUTO;
  bool beam_w = true;          // split scorer on the one-wave-per-SIMD kernel (beam_kernel_w.hip.inc); DM_BEAM_W=0 in the environment selects the LDS-fed kernel
  bool split_dirty = true;
  void *d_wsplit = nullptr;
  double jtm_score_s = 0, jtm_rebal_s = 0;   // dm_jtm_last_step_seconds
  void *d_rows_split = nullptr;   // general-rows split kernel: fp16 hi / lo planes of W1a and M = W1b att.W, then M in fp32
  int sh_r = 0; bool rows_split_dirty = true;
  bool emb_split_dirty = true;   // the pre-split copy of the table lags the scales (ensure_split_scales / ensure_split)
  bool call_split = false;       // scorer arithmetic of the search being planned (split_for_call)
  // incremental refresh inside a training loop (ensure_split_scales): the table changed only in the ACTIVE rows of the Adam step
  bool table_dense_change = true;   // ... unless something rewrote it wholesale since the last full scan (load, dense Adam step, f64 -> f32 mirror)
  bool sh_e_valid = false;          // sh_e comes from a full scan of the current table lineage
  bool emb_split_need_full = true;  // d_emb_split is not (scale sh_e, stale in active rows only)
  bool emb_split_patch = false;
  unsigned long long active_rows_host = 0;   // length of the active-row list at the last Adam step
  void *d_emb_split = nullptr;     // pre-split table of the W kernel (beam_kernel_w.hip.inc)
  size_t emb_split_bytes = 0;
  unsigned *d_maxabs = nullptr;
  int sh_e = 0, sh_w = 0;
  void *d_att_wT_t = nullptr, *d_l1T_t = nullptr;  // transposes in the loaded dtype (general forward)
  // fp64 beam kernel (beam_kernel_f64.hip.inc): A / B fragments of att.W, W1a, W1b; per-team K / G fragment scratch
  void *d_frag64 = nullptr, *d_scratch64 = nullptr;
  bool frag64_dirty = true;
  double b2_64 = 0.0;            // l2.b of the f64 model (read back when the fragments are built)
  size_t scratch64_bytes = 0;
  // training state (dm_train_init)
  bool train_ready = false;
  dm_adam_opts adam{};
  int adam_t = 0;
  void *d_grad = nullptr, *d_adam_s = nullptr, *d_adam_r = nullptr, *d_loss = nullptr;   // in the loaded dtype
  void *d_tr64 = nullptr;        // f64 model: A fragments of att.W, W1a, W1b and of their transposes for the training kernels
  void *d_tail32 = nullptr;      // f64 model: f32 copy of the small matrices (source of the f32 mirror's fragments)
  bool f32_mirror_dirty = false; // f64 model whose weights moved: the f32 copies the throughput beam kernels read are stale
  double last_loss = 0.0;
  f32x4 *d_attTA = nullptr, *d_w1aTA = nullptr, *d_w1bTA = nullptr;
  unsigned *d_touch_bits = nullptr;
  int32_t *d_touch_list = nullptr;
  unsigned long long *d_touch_cnt = nullptr;
  unsigned *d_active_bits = nullptr;   // rows a gradient has ever reached since dm_train_init (the rows the Adam step has to visit)
  int32_t *d_active_list = nullptr;
  unsigned long long *d_active_cnt = nullptr;
  int adam_last_sparse = 0; unsigned long long adam_last_rows = 0;
  size_t touch_cap = 0, touch_ub = 0;   // list capacity; host-side upper bound of its length since the last Adam step
  // measurement
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  std::vector<int> ev_kind;    // per pair: 0 = a main kernel, 1 = the second pass over the users the one-wave kernel deferred
  int ev_next_kind = 0;
  bool ev_skip = false;        // the single-request path launches without its event pair ...
  bool time_direct = false;    // ... unless DM_TIME_DIRECT=1 asks for kernel timings of that path too (probes)
  bool direct_ok = true;       // host-mapped single-request path enabled (DM_NO_DIRECT=1 turns it off)
  char last_kernel[64] = "";   // the search kernel of the last beam search (measurement: dm_last_beam_kernel)
  unsigned long long *d_rows = nullptr;
  unsigned long long *d_phase = nullptr;   // 8 debug counters
  int64_t last_rows = 0;
  // Deep-Retrieval model (dm_dr_load_model)
  dm_dr_state *dr = nullptr;
  // request arena of the host-buffer entry points (grow only: no hipMalloc / hipFree on the request path)
  void *d_req = nullptr;
  size_t req_bytes = 0;
  unsigned long long h_rows = 0;
  char *h_stage = nullptr;     // pinned staging block for small host-buffer requests (one upload, one download per call)
  char *d_stage = nullptr;     // the same block as the kernels address it (hipHostGetDevicePointer): single-request path
  size_t stage_bytes = 0;
  // pipelined host-buffer searches (host_pipe_*): a second, non-blocking stream for the result downloads + one event per chunk
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> chunk_ev;
  bool rows_keep = false;      // a later chunk of one request: the scored-rows counter keeps counting
  // cached search workspace
  void *d_ws = nullptr;
  size_t ws_bytes = 0;
  // device-side negative sampler (sampler.hip.inc): per-level code / cumulative-probability tables, per-call scratch
  int32_t *d_lv_codes = nullptr;
  double *d_lv_cdf = nullptr;
  int64_t *d_lv_start = nullptr;
  void *d_samp = nullptr;
  size_t samp_bytes = 0;
  // multi-GPU exchange (comm.hip.inc): the attached communicator (not owned) and the staging area of dm_train_sync_gradients
  void *d_defer = nullptr;       // users the W kernel hands to the LDS-fed kernel: [count u64 | queue head u64 | ids]
  size_t defer_bytes = 0;
  // JTM: the catalogue's training rows kept on the device across gap steps (dm_jtm_cache_rows)
  int64_t *d_jtm_off = nullptr;
  int32_t *d_jtm_ritem = nullptr, *d_jtm_rids = nullptr;
  int32_t *d_jtm_rseq = nullptr;       // the rows' history CODES [rows][L] and pad masks [rows] (built once per cached catalogue and id map:
  unsigned *d_jtm_rmask = nullptr;     //  they do not change between the gap steps unless the ancestors are taken per level — hierarchical mode)
  uint64_t jtm_rseq_ids_epoch = 0;     // id-map generation the codes were built from
  int64_t jtm_rseq_num_index = 0;      // ... and the table size they were bounds-checked against
  std::vector<int64_t> jtm_off;
  int64_t jtm_i_lo = 0, jtm_i_hi = 0, jtm_R_base = 0;      // the items whose rows are on this device (dm_jtm_cache_rows_range) and their first row
  int jtm_L = 0;
  struct dm_comm *comm = nullptr;
  dm_sync_stats sync_stats{};
  dm_jtm_stats jtm_stats{};
  dm_otm_stats otm_stats{};
  void *d_sync = nullptr;
  size_t sync_bytes = 0;
};

static std::string g_create_err;

#define HIPCHK(h, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      char b_[512];                                                                       \
      snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (h)->err = b_;                                                                      \
      return DM_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)

static int fail(dm_ctx *h, int code, const std::string &msg) {
  if (h) h->err = msg; else g_create_err = msg;
  return code;
}

// ------------------------------------------------------------ small kernels
__global__ void dm_f64_to_f32_kernel(const double *in, float *out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (float)in[i];
}

__global__ void dm_pad_rowmask_kernel(const int32_t *pad_flat, int64_t n_pad, int L, unsigned *rowmask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) {
    int64_t idx = pad_flat[i];
    atomicOr(&rowmask[idx / L], 1u << (idx % L));
  }
}

// General DIN forward, one wave per row, any (code, history, mask) per row:
// Module.forward of tdm/.../model/DIN.scala:18-42 for arbitrary batches (the beam kernels
// use the per-user restructured form instead).  T = float | double.
template <typename T>
struct DinFwdParams {
  const T *emb, *att_wT, *l1T, *b1, *w2;  // att_wT [E k][E o], l1T [2E k][E o]
  T b2;
  int E, L;
  int64_t B;
  const int32_t *codes, *seqs;
  const unsigned *rowmask;
  T *out;
  T sm_scale;                  // 1 / sqrt(embedSize) of the model (the table may be zero-padded to E)
};

template <typename T>
__device__ __forceinline__ T dm_wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void dm_din_forward_kernel(DinFwdParams<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem_fw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int E = p.E, L = p.L;
  T *q = (T *)smem_fw + (size_t)wave * (3 * E + 32);
  T *comb = q + E, *att = comb + E, *sc = att + E;
  const T scale = p.sm_scale;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.B; row += (int64_t)gridDim.x * 4) {
    const int32_t code = p.codes[row];
    for (int e = lane; e < E; e += 64) q[e] = code >= 0 ? p.emb[(int64_t)code * E + e] : (T)0;
    __builtin_amdgcn_wave_barrier();
    const unsigned mask = p.rowmask ? p.rowmask[row] : 0u;
    for (int j = 0; j < L; j++) {
      const int32_t c = p.seqs[row * L + j];
      T part = 0;
      if (c >= 0)
        for (int e = lane; e < E; e += 64) part += q[e] * p.emb[(int64_t)c * E + e];
      T s = dm_wave_sum(part) * scale;
      if ((mask >> j) & 1u) s = (T)(-FLT_MAX);
      if (lane == 0) sc[j] = s;
    }
    __builtin_amdgcn_wave_barrier();
    T mx = sc[0];
    for (int j = 1; j < L; j++) mx = sc[j] > mx ? sc[j] : mx;
    T sum;
    sum = 0;
    for (int j = 0; j < L; j++) {
      T e = sizeof(T) == 4 ? (T)expf((float)(sc[j] - mx)) : (T)exp((double)(sc[j] - mx));
      sum += e;
    }
    const T inv = (T)1 / sum;
    for (int e = lane; e < E; e += 64) comb[e] = 0;
    for (int j = 0; j < L; j++) {
      const int32_t c = p.seqs[row * L + j];
      T ex = sizeof(T) == 4 ? (T)expf((float)(sc[j] - mx)) : (T)exp((double)(sc[j] - mx));
      T pj = ex * inv;
      if (c >= 0)
        for (int e = lane; e < E; e += 64) comb[e] += pj * p.emb[(int64_t)c * E + e];
    }
    __builtin_amdgcn_wave_barrier();
    for (int o = lane; o < E; o += 64) {
      T a = 0;
      for (int k = 0; k < E; k++) a += comb[k] * p.att_wT[(size_t)k * E + o];
      att[o] = a;
    }
    __builtin_amdgcn_wave_barrier();
    T part = 0;
    for (int o = lane; o < E; o += 64) {
      T hsum = 0;
      for (int k = 0; k < E; k++) hsum += q[k] * p.l1T[(size_t)k * E + o];
      for (int k = 0; k < E; k++) hsum += att[k] * p.l1T[(size_t)(E + k) * E + o];
      hsum += p.b1[o];
      hsum = hsum > 0 ? hsum : (T)0;
      part += hsum * p.w2[o];
    }
    T logit = dm_wave_sum(part) + p.b2;
    if (lane == 0) p.out[row] = logit;
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------ helpers
static int dm_alloc(dm_ctx *h, void **p, size_t bytes) {
  HIPCHK(h, hipMalloc(p, bytes ? bytes : 16));
  return DM_OK;
}
#define ALLOC(h, ptr, bytes)                                   \
  do {                                                         \
    int rc_ = dm_alloc((h), (void **)&(ptr), (bytes));         \
    if (rc_ != DM_OK) return rc_;                              \
  } while (0)
static void dm_free_ptr(void *p) { if (p) (void)hipFree(p); }

// The 256 KB pinned staging block of the small-request paths.  It is also what the single-request kernels read and write in place
// and whose count words the host polls, so it is allocated COHERENT explicitly (fine-grained: device stores become visible to the
// host without a kernel boundary) and the kernels get the address hipHostGetDevicePointer reports, not the host pointer.
static int ensure_stage(dm_ctx *h) {
  if (h->h_stage && h->stage_bytes >= (256u << 10)) return DM_OK;
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  h->h_stage = nullptr; h->d_stage = nullptr; h->stage_bytes = 0;
  if (hipHostMalloc((void **)&h->h_stage, 256u << 10, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess)
    return fail(h, DM_ERR_HIP, "hipHostMalloc failed");
  void *dp = nullptr;
  if (hipHostGetDevicePointer(&dp, h->h_stage, 0) != hipSuccess || !dp) {
    (void)hipHostFree(h->h_stage); h->h_stage = nullptr;
    return fail(h, DM_ERR_HIP, "hipHostGetDevicePointer failed");
  }
  h->d_stage = (char *)dp; h->stage_bytes = 256u << 10;
  return DM_OK;
}

// Mask.scala:10 — scale = 1 / sqrt(embedSize) of the model as loaded (zero padding of the table does not change it)
static double sm_scale64(const dm_ctx *h) { return 1.0 / sqrt((double)(h->embed_log > 0 ? h->embed_log : h->embed)); }
static float sm_scale32(const dm_ctx *h) { return (float)sm_scale64(h); }

static int level_start_int(int candidate_num, int *start, int *level) {
  if (candidate_num <= 0) return DM_ERR_INVALID;
  int lv = 0;
  while ((2 << lv) <= candidate_num) lv++;   // floor(log2)
  *level = lv;
  *start = (1 << lv) - 1;
  return DM_OK;
}

// ------------------------------------------------------------------ C ABI
// (every dm_* below is declared extern "C" by include/dismember_hip.h)

int dm_version(void) { return DM_VERSION; }

int dm_device_count(int *count) {
  if (!count) return DM_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; g_create_err = hipGetErrorString(e); return DM_ERR_HIP; }
  *count = n;
  return DM_OK;
}

const char *dm_last_error(dm_handle_t h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int dm_create(int device_id, dm_handle_t *out) {
  if (!out) return fail(nullptr, DM_ERR_INVALID, "dm_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, DM_ERR_HIP, std::string("dm_create: no HIP device available (") +
                                         (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                                         "); this library has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(nullptr, DM_ERR_INVALID, "dm_create: device_id out of range");
  dm_ctx *h = new dm_ctx();
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) { delete h; return fail(nullptr, DM_ERR_HIP, "hipSetDevice failed"); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete h; return fail(nullptr, DM_ERR_HIP, "hipGetDeviceProperties failed"); }
  h->n_cu = prop.multiProcessorCount;
  { const char *e_ = getenv("DM_BEAM_W"); if (e_) h->beam_w = e_[0] == '1'; }
  { const char *e_ = getenv("DM_NO_DIRECT"); if (e_ && e_[0] == '1') h->direct_ok = false; }
  { const char *e_ = getenv("DM_TIME_DIRECT"); if (e_ && e_[0] == '1') h->time_direct = true; }
  if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return fail(nullptr, DM_ERR_HIP, "hipStreamCreate failed"); }
  if (hipMalloc((void **)&h->d_rows, 64) != hipSuccess) { delete h; return fail(nullptr, DM_ERR_HIP, "hipMalloc failed"); }
  (void)hipMemset(h->d_rows, 0, 64);
  if (hipMalloc((void **)&h->d_phase, 128) == hipSuccess) (void)hipMemset(h->d_phase, 0, 128);
  *out = h;
  return DM_OK;
}

static inline void model_changed(dm_ctx *h) { h->model_epoch.fetch_add(1); }
static int clone_enter(dm_ctx *c);
// entry points that replace the model or train it: the owning handle only
#define DM_OWNER_ONLY(h, who) do { if ((h)->parent) return fail((h), DM_ERR_STATE, who ": not on a clone (dm_clone) - load and train through the owning handle"); } while (0)
// read-only entry points: a clone first brings its mirror of the parent's model up to date
#define DM_CLONE_ENTER(h) do { if ((h)->parent) { const int rc_ce_ = clone_enter(h); if (rc_ce_ != DM_OK) return rc_ce_; } } while (0)

static void free_tree(dm_ctx *h) {
  model_changed(h);
  dm_free_ptr(h->d_lv_codes); dm_free_ptr(h->d_lv_cdf); dm_free_ptr(h->d_lv_start);
  h->d_lv_codes = nullptr; h->d_lv_cdf = nullptr; h->d_lv_start = nullptr;
  dm_free_ptr(h->d_exists); dm_free_ptr(h->d_leaf); dm_free_ptr(h->d_node_id); dm_free_ptr(h->d_leaf_codes);
  h->d_exists = h->d_leaf = nullptr; h->d_node_id = h->d_leaf_codes = nullptr; h->tree_loaded = false;
}
static void free_weights(dm_ctx *h) {
  model_changed(h);
  if (h->emb32_owned) dm_free_ptr(h->d_emb32);
  dm_free_ptr(h->d_compact); dm_free_ptr(h->d_wfrag); dm_free_ptr(h->d_afrag); dm_free_ptr(h->d_bfrag); dm_free_ptr(h->d_attA); dm_free_ptr(h->d_w1aA); dm_free_ptr(h->d_w1bA);
  dm_free_ptr(h->d_b1); dm_free_ptr(h->d_w2); dm_free_ptr(h->d_att_wT_t); dm_free_ptr(h->d_l1T_t);
  dm_free_ptr(h->d_wsplit); dm_free_ptr(h->d_maxabs); h->d_wsplit = nullptr; h->d_maxabs = nullptr; h->split_dirty = true; h->table_dense_change = true; h->sh_e_valid = false; h->emb_split_need_full = true;
  dm_free_ptr(h->d_rows_split); h->d_rows_split = nullptr; h->rows_split_dirty = true;
  dm_free_ptr(h->d_emb_split); h->d_emb_split = nullptr; h->emb_split_bytes = 0;
  dm_free_ptr(h->d_frag64); h->d_frag64 = nullptr; h->frag64_dirty = true;
  dm_free_ptr(h->d_tr64); h->d_tr64 = nullptr; dm_free_ptr(h->d_tail32); h->d_tail32 = nullptr; h->f32_mirror_dirty = false;
  h->d_compact = nullptr; h->d_emb32 = nullptr; h->emb32_owned = false; h->d_wfrag = nullptr;
  dm_free_ptr(h->d_grad); dm_free_ptr(h->d_adam_s); dm_free_ptr(h->d_adam_r); dm_free_ptr(h->d_loss); dm_free_ptr(h->d_attTA);
  dm_free_ptr(h->d_w1aTA); dm_free_ptr(h->d_w1bTA); dm_free_ptr(h->d_touch_bits); dm_free_ptr(h->d_touch_list); dm_free_ptr(h->d_touch_cnt);
  dm_free_ptr(h->d_active_bits); dm_free_ptr(h->d_active_list); dm_free_ptr(h->d_active_cnt); h->d_active_bits = nullptr; h->d_active_list = nullptr; h->d_active_cnt = nullptr;
  h->d_grad = h->d_adam_s = h->d_adam_r = h->d_loss = nullptr; h->d_attTA = h->d_w1aTA = h->d_w1bTA = nullptr;
  h->d_touch_bits = nullptr; h->d_touch_list = nullptr; h->d_touch_cnt = nullptr; h->train_ready = false; h->touch_cap = 0; h->touch_ub = 0;
  h->d_afrag = h->d_bfrag = nullptr; h->d_attA = h->d_w1aA = h->d_w1bA = nullptr; h->d_b1 = h->d_w2 = nullptr; h->d_att_wT_t = h->d_l1T_t = nullptr; h->w_loaded = false;
}

// Everything a clone mirrors from its parent: the tree, the id maps, the weights and every derived copy (fragment orders, split planes,
// the pre-split table, the f32 mirror of an f64 model, the fp64 fragments).  One list drives the mirror and the clone's tear-down.
#define DM_SHARED_FIELDS(X)                                                                                                              \
  X(tree_loaded) X(ids_loaded) X(leaves_at_max_only) X(max_level) X(n_slots) X(n_leaf_nodes) X(d_exists) X(d_leaf) X(d_node_id)          \
  X(d_id_to_code) X(d_leaf_codes) X(non_leaf_offset) X(max_code) X(w_loaded) X(dtype) X(embed) X(embed_log) X(num_index) X(d_compact)    \
  X(d_emb32) X(d_wfrag) X(d_afrag) X(d_bfrag) X(d_attA) X(d_w1aA) X(d_w1bA) X(d_b1) X(d_w2) X(b2) X(d_wsplit) X(d_rows_split) X(sh_r)    \
  X(d_emb_split) X(emb_split_bytes) X(d_maxabs) X(sh_e) X(sh_w) X(d_att_wT_t) X(d_l1T_t) X(d_frag64) X(b2_64) X(d_tail32)                \
  X(d_lv_codes) X(d_lv_cdf) X(d_lv_start)

static void clone_mirror(dm_ctx *c, dm_ctx *p) {
#define X(f) c->f = p->f;
  DM_SHARED_FIELDS(X)
#undef X
  if (c->seen_ids_epoch != p->ids_epoch || c->h_exists.size() != p->h_exists.size()) {      // (host vectors: O(catalogue), copied when the tree / id maps changed, not after every Adam step)
    c->h_id_to_code = p->h_id_to_code;
    c->h_exists = p->h_exists;
    c->seen_ids_epoch = p->ids_epoch;
  }
  c->emb32_owned = false;
  // the parent's copies were brought up to date before the mirror was taken: nothing is stale for the clone, and nothing is rebuilt by it
  c->split_dirty = false; c->emb_split_dirty = false; c->rows_split_dirty = false; c->frag64_dirty = false; c->f32_mirror_dirty = false;
  c->table_dense_change = false; c->sh_e_valid = p->sh_e_valid; c->emb_split_need_full = false; c->emb_split_patch = false;
  c->seen_epoch = p->model_epoch.load();
}
static void clone_forget(dm_ctx *c) {         // the clone owns none of it
  dm_ctx z;
#define X(f) c->f = z.f;
  DM_SHARED_FIELDS(X)
#undef X
  c->emb32_owned = false;
}

static bool use_split(const dm_ctx *h);
static bool use_f64_beam(const dm_ctx *h);
static int ensure_split(dm_ctx *h);
static int ensure_rows_split(dm_ctx *h);
static int ensure_f32_mirror(dm_ctx *h);
static int ensure_frags64(dm_ctx *h);

// A clone's read-only entry points start here.  Fast path: the parent's model has not changed since the mirror was taken (one atomic
// load).  Otherwise, under the parent's lock: the copies this clone's scorer needs are brought up to date ON THE PARENT (its stream,
// its buffers; no-ops when clean), the parent's stream is drained, and the mirror is retaken.  Weight updates through the parent must
// not overlap a clone's search in flight — the reference's workers wait for each other the same way (LocalOptimizer.scala:73-80).
static int clone_enter(dm_ctx *c) {
  dm_ctx *p = c->parent;
  if (c->seen_epoch == p->model_epoch.load()) return DM_OK;
  std::lock_guard<std::recursive_mutex> lock(p->mu);
  if (hipSetDevice(p->device) != hipSuccess) return fail(c, DM_ERR_HIP, "dm_clone: hipSetDevice failed");
  for (int pass = 0; pass < 2; pass++) {
    clone_mirror(c, p);                         // (first pass: the model's shape, so that the scorer choice below is this clone's)
    if (!p->w_loaded) break;
    int rc = DM_OK;
    if (use_f64_beam(c)) rc = ensure_frags64(p);        // (decided on the clone: the parent's model with the clone's own scorer setting)
    else {
      rc = ensure_f32_mirror(p);
      if (rc == DM_OK && use_split(c)) { rc = ensure_split(p); if (rc == DM_OK) rc = ensure_rows_split(p); }
    }
    if (rc != DM_OK) { c->err = p->err; return rc; }
    if (hipStreamSynchronize(p->stream) != hipSuccess) return fail(c, DM_ERR_HIP, "dm_clone: the parent's stream failed");
  }
  return DM_OK;
}

int dm_clone(dm_handle_t h, dm_handle_t *out) {
  if (!h || !out) return DM_ERR_INVALID;
  *out = nullptr;
  dm_ctx *root = h->parent ? h->parent : h;      // a clone of a clone shares the same owner
  dm_handle_t c = nullptr;
  int rc = dm_create(root->device, &c);
  if (rc != DM_OK) return fail(h, rc, g_create_err);
  c->parent = root;
  c->scorer_mode = h->scorer_mode;
  root->n_clones.fetch_add(1);
  rc = clone_enter(c);
  if (rc != DM_OK) { h->err = c->err; root->n_clones.fetch_sub(1); c->parent = nullptr; dm_destroy(c); return rc; }
  *out = c;
  return DM_OK;
}

int dm_destroy(dm_handle_t h) {
  if (!h) return DM_ERR_INVALID;
  if (h->n_clones.load() > 0) return fail(h, DM_ERR_STATE, "dm_destroy: the handle still has clones (dm_clone): destroy them first");
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->parent) { clone_forget(h); h->parent->n_clones.fetch_sub(1); h->parent = nullptr; }
  free_tree(h); free_weights(h); dm_dr_free(h->dr);
  dm_free_ptr(h->d_id_to_code); dm_free_ptr(h->d_rows); dm_free_ptr(h->d_phase); dm_free_ptr(h->d_ws); dm_free_ptr(h->d_req); dm_free_ptr(h->d_sync);
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  dm_free_ptr(h->d_lv_codes); dm_free_ptr(h->d_lv_cdf); dm_free_ptr(h->d_lv_start); dm_free_ptr(h->d_samp); dm_free_ptr(h->d_defer); dm_free_ptr(h->d_scratch64);
  dm_free_ptr(h->d_jtm_off); dm_free_ptr(h->d_jtm_ritem); dm_free_ptr(h->d_jtm_rids); dm_free_ptr(h->d_jtm_rseq); dm_free_ptr(h->d_jtm_rmask);
  for (auto &pr : h->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto &e_ : h->chunk_ev) (void)hipEventDestroy(e_);
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return DM_OK;
}

int dm_synchronize(dm_handle_t h) {
  if (!h) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DM_OK;
}

int dm_level_start(int candidate_num, int *start_code, int *level) {
  if (!start_code || !level) return DM_ERR_INVALID;
  return level_start_int(candidate_num, start_code, level);
}

int dm_load_tree_tdm(dm_handle_t h, const int32_t *codes, const int32_t *node_ids, const uint8_t *is_leaf,
                     int64_t n_nodes, int max_level) {
  if (!h) return DM_ERR_INVALID;
  DM_OWNER_ONLY(h, "dm_load_tree_tdm");
  if (!codes || !node_ids || !is_leaf || n_nodes <= 0 || max_level < 0 || max_level > 30)
    return fail(h, DM_ERR_INVALID, "dm_load_tree_tdm: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  int64_t mc = -1;
  for (int64_t i = 0; i < n_nodes; i++) {
    if (codes[i] < 0) return fail(h, DM_ERR_INVALID, "dm_load_tree_tdm: negative code");
    if (codes[i] > mc) mc = codes[i];
  }
  const int64_t n_slots = mc + 1;
  const int64_t words = (n_slots + 31) / 32 + 1;
  std::vector<uint32_t> ex(words, 0), lf(words, 0);
  std::vector<int32_t> nid(n_slots, 0), leaf_codes;
  bool at_max_only = true;
  const int64_t first_max = ((int64_t)1 << max_level) - 1;
  for (int64_t i = 0; i < n_nodes; i++) {
    int64_t c = codes[i];
    ex[c >> 5] |= 1u << (c & 31);
    nid[c] = node_ids[i];
    if (is_leaf[i]) { lf[c >> 5] |= 1u << (c & 31); leaf_codes.push_back((int32_t)c); if (c < first_max) at_max_only = false; }
    else if (c >= first_max) at_max_only = false;   // a non-leaf on the last level: take the general path
  }
  free_tree(h);
  ALLOC(h, h->d_exists, words * 4);
  ALLOC(h, h->d_leaf, words * 4);
  ALLOC(h, h->d_node_id, n_slots * 4);
  ALLOC(h, h->d_leaf_codes, leaf_codes.size() * 4);
  HIPCHK(h, hipMemcpy(h->d_exists, ex.data(), words * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_leaf, lf.data(), words * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_node_id, nid.data(), n_slots * 4, hipMemcpyHostToDevice));
  if (!leaf_codes.empty())
    HIPCHK(h, hipMemcpy(h->d_leaf_codes, leaf_codes.data(), leaf_codes.size() * 4, hipMemcpyHostToDevice));
  h->h_exists = ex;
  h->ids_epoch++;              // (clones re-copy their host vectors; per-row code caches built against the old tree are dropped)
  h->n_slots = n_slots; h->n_leaf_nodes = (int64_t)leaf_codes.size(); h->max_level = max_level;
  h->leaves_at_max_only = at_max_only; h->tree_loaded = true;
  return DM_OK;
}

int dm_load_id_maps(dm_handle_t h, const int32_t *leaf_item_ids, const int32_t *leaf_codes, int64_t n) {
  if (!h) return DM_ERR_INVALID;
  DM_OWNER_ONLY(h, "dm_load_id_maps");
  if (!leaf_item_ids || !leaf_codes || n <= 0) return fail(h, DM_ERR_INVALID, "dm_load_id_maps: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  int32_t mid = -1, mcode = -1;
  for (int64_t i = 0; i < n; i++) {
    if (leaf_item_ids[i] > mid) mid = leaf_item_ids[i];
    if (leaf_codes[i] > mcode) mcode = leaf_codes[i];
  }
  if (mid < 0) return fail(h, DM_ERR_INVALID, "dm_load_id_maps: no non-negative item id");
  model_changed(h);
  h->ids_epoch++;
  h->non_leaf_offset = mid + 1;   // DistTree.scala:35
  h->max_code = mcode;            // DistTree.scala:36
  h->h_id_to_code.assign((size_t)h->non_leaf_offset, -1);
  for (int64_t i = 0; i < n; i++)
    if (leaf_item_ids[i] >= 0) h->h_id_to_code[leaf_item_ids[i]] = leaf_codes[i];
  dm_free_ptr(h->d_id_to_code); h->d_id_to_code = nullptr;
  ALLOC(h, h->d_id_to_code, h->h_id_to_code.size() * 4);
  HIPCHK(h, hipMemcpy(h->d_id_to_code, h->h_id_to_code.data(), h->h_id_to_code.size() * 4, hipMemcpyHostToDevice));
  h->ids_loaded = true;
  return DM_OK;
}

int dm_tdm_id_to_code(dm_handle_t h, const int32_t *item_ids, int n, int32_t *codes, int32_t *mask_pos, int *n_mask) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  if (!h->ids_loaded) return fail(h, DM_ERR_STATE, "dm_tdm_id_to_code: id maps not loaded");
  if (!item_ids || !codes || !mask_pos || !n_mask || n < 0) return fail(h, DM_ERR_INVALID, "dm_tdm_id_to_code: bad arguments");
  int nm = 0;
  for (int i = 0; i < n; i++) {
    int32_t id = item_ids[i];
    if (id == 0) { mask_pos[nm++] = i; codes[i] = -1; }
    else if (id < h->non_leaf_offset && id >= 0 && h->h_id_to_code[id] >= 0) codes[i] = h->h_id_to_code[id];
    else {
      int32_t t = (int32_t)((uint32_t)id - (uint32_t)h->non_leaf_offset);
      if (t > h->max_code) { mask_pos[nm++] = i; codes[i] = -1; } else codes[i] = t;
    }
  }
  *n_mask = nm;
  return DM_OK;
}

// derived, fragment-ordered copies of the small matrices (att.W, l1.W, l1.b, l2.W, l2.b) from the
// host copy of the tail of the compact vector
template <typename T>
static int upload_derived(dm_ctx *h, int E, const T *att_w) {
  const T *l1_w = att_w + (int64_t)E * E, *l1_b = l1_w + (int64_t)E * 2 * E;
  const T *l2_w = l1_b + E, *l2_b = l2_w + E;
  const int NJ = E / 16, NT = E / 16;
  std::vector<float> wfrag((size_t)NJ * NT * 64 * 4), afrag(wfrag.size()), bfrag(wfrag.size()), b1(E), w2(E);
  for (int jc = 0; jc < NJ; jc++)
    for (int nt = 0; nt < NT; nt++)
      for (int ln = 0; ln < 64; ln++)
        for (int t = 0; t < 4; t++) {
          int g = ln >> 4, n = ln & 15;
          const size_t fi = (((size_t)jc * NT + nt) * 64 + ln) * 4 + t;
          const int k = 16 * jc + 4 * g + t, o = 16 * nt + n;
          wfrag[fi] = (float)l1_w[(size_t)o * 2 * E + k];        // B[k][o] = W1a[o][k]
          afrag[fi] = (float)att_w[(size_t)o * E + k];            // B[k][o] = att_w[o][k]
          bfrag[fi] = (float)l1_w[(size_t)o * 2 * E + E + k];    // B[k][o] = W1b[o][k]
        }
  for (int o = 0; o < E; o++) { b1[o] = (float)l1_b[o]; w2[o] = (float)l2_w[o]; }
  h->b2 = (float)l2_b[0];
  // A-fragment order for the transposed products of the rows kernel: [mt][jc][lane] float4,
  // element t = W[16mt + (lane&15)][16jc + 4(lane>>4) + t]
  std::vector<float> attA(wfrag.size()), w1aA(wfrag.size()), w1bA(wfrag.size());
  for (int mt = 0; mt < NT; mt++)
    for (int jc = 0; jc < NJ; jc++)
      for (int ln = 0; ln < 64; ln++)
        for (int t = 0; t < 4; t++) {
          const size_t fi = (((size_t)mt * NJ + jc) * 64 + ln) * 4 + t;
          const int o = 16 * mt + (ln & 15), k = 16 * jc + 4 * (ln >> 4) + t;
          attA[fi] = (float)att_w[(size_t)o * E + k];
          w1aA[fi] = (float)l1_w[(size_t)o * 2 * E + k];
          w1bA[fi] = (float)l1_w[(size_t)o * 2 * E + E + k];
        }
  ALLOC(h, h->d_attA, attA.size() * 4);
  ALLOC(h, h->d_w1aA, w1aA.size() * 4);
  ALLOC(h, h->d_w1bA, w1bA.size() * 4);
  HIPCHK(h, hipMemcpy(h->d_attA, attA.data(), attA.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_w1aA, w1aA.data(), w1aA.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_w1bA, w1bA.data(), w1bA.size() * 4, hipMemcpyHostToDevice));
  ALLOC(h, h->d_wfrag, wfrag.size() * 4);
  ALLOC(h, h->d_afrag, afrag.size() * 4);
  ALLOC(h, h->d_bfrag, bfrag.size() * 4);
  ALLOC(h, h->d_b1, E * 4);
  ALLOC(h, h->d_w2, E * 4);
  HIPCHK(h, hipMemcpy(h->d_wfrag, wfrag.data(), wfrag.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_afrag, afrag.data(), afrag.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_bfrag, bfrag.data(), bfrag.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_b1, b1.data(), E * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_w2, w2.data(), E * 4, hipMemcpyHostToDevice));
  // transposes in the loaded dtype for the general forward
  std::vector<T> attT_t((size_t)E * E), l1T_t((size_t)2 * E * E);
  for (int o = 0; o < E; o++) {
    for (int k = 0; k < E; k++) attT_t[(size_t)k * E + o] = att_w[(size_t)o * E + k];
    for (int k = 0; k < 2 * E; k++) l1T_t[(size_t)k * E + o] = l1_w[(size_t)o * 2 * E + k];
  }
  ALLOC(h, h->d_att_wT_t, attT_t.size() * sizeof(T));
  ALLOC(h, h->d_l1T_t, l1T_t.size() * sizeof(T));
  HIPCHK(h, hipMemcpy(h->d_att_wT_t, attT_t.data(), attT_t.size() * sizeof(T), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_l1T_t, l1T_t.data(), l1T_t.size() * sizeof(T), hipMemcpyHostToDevice));
  return DM_OK;
}

// Embedding sizes the kernels are built for; any other size up to 128 is zero-padded to the next one at load (S/nn/Attention.scala,
// T/model/DIN.scala take any embedSize).  Zero columns / rows change no product: q.k, att.W k, W1 [q ; att] and w2 . h keep their
// values, the padded hidden units are relu(0) = 0, and training leaves the padding at zero (its gradients are products with zero
// inputs; Adam moves a parameter whose gradient history is zero by 0 / (0 + eps)).  Only the softmax scale 1 / sqrt(embedSize)
// (Mask.scala:10) must stay the model's: sm_scale64().
static int native_embed(int E) { return E <= 0 ? 0 : E <= 16 ? 16 : E <= 32 ? 32 : E <= 64 ? 64 : E <= 128 ? 128 : 0; }
static int64_t compact_len_for(int64_t num_index, int64_t E) { return num_index * E + 3 * E * E + 2 * E + 1; }

template <typename T>
static void pad_compact(const T *src, int E, int Ep, int64_t NI, T *dst) {       // dst: compact_len_for(NI, Ep) elements, zeroed here
  memset(dst, 0, (size_t)compact_len_for(NI, Ep) * sizeof(T));
  for (int64_t r = 0; r < NI; r++) memcpy(dst + r * Ep, src + r * E, (size_t)E * sizeof(T));
  const T *s_att = src + NI * E, *s_l1 = s_att + (int64_t)E * E, *s_b1 = s_l1 + (int64_t)E * 2 * E, *s_w2 = s_b1 + E;
  T *d_att = dst + NI * Ep, *d_l1 = d_att + (int64_t)Ep * Ep, *d_b1 = d_l1 + (int64_t)Ep * 2 * Ep, *d_w2 = d_b1 + Ep;
  for (int o = 0; o < E; o++) {
    memcpy(d_att + (int64_t)o * Ep, s_att + (int64_t)o * E, (size_t)E * sizeof(T));
    memcpy(d_l1 + (int64_t)o * 2 * Ep, s_l1 + (int64_t)o * 2 * E, (size_t)E * sizeof(T));                  // W1a half
    memcpy(d_l1 + (int64_t)o * 2 * Ep + Ep, s_l1 + (int64_t)o * 2 * E + E, (size_t)E * sizeof(T));        // W1b half
  }
  memcpy(d_b1, s_b1, (size_t)E * sizeof(T));
  memcpy(d_w2, s_w2, (size_t)E * sizeof(T));
  d_w2[Ep] = s_w2[E];
}
template <typename T>
static void unpad_compact(const T *src, int E, int Ep, int64_t NI, T *dst) {     // src padded, dst: compact_len_for(NI, E) elements
  for (int64_t r = 0; r < NI; r++) memcpy(dst + r * E, src + r * Ep, (size_t)E * sizeof(T));
  const T *s_att = src + NI * Ep, *s_l1 = s_att + (int64_t)Ep * Ep, *s_b1 = s_l1 + (int64_t)Ep * 2 * Ep, *s_w2 = s_b1 + Ep;
  T *d_att = dst + NI * E, *d_l1 = d_att + (int64_t)E * E, *d_b1 = d_l1 + (int64_t)E * 2 * E, *d_w2 = d_b1 + E;
  for (int o = 0; o < E; o++) {
    memcpy(d_att + (int64_t)o * E, s_att + (int64_t)o * Ep, (size_t)E * sizeof(T));
    memcpy(d_l1 + (int64_t)o * 2 * E, s_l1 + (int64_t)o * 2 * Ep, (size_t)E * sizeof(T));
    memcpy(d_l1 + (int64_t)o * 2 * E + E, s_l1 + (int64_t)o * 2 * Ep + Ep, (size_t)E * sizeof(T));
  }
  memcpy(d_b1, s_b1, (size_t)E * sizeof(T));
  memcpy(d_w2, s_w2, (size_t)E * sizeof(T));
  d_w2[E] = s_w2[Ep];
}

template <typename T>
static int load_weights_t(dm_ctx *h, int E, int64_t num_index, const T *w, int64_t n_elems) {
  const int64_t need = num_index * E + (int64_t)E * E + (int64_t)E * 2 * E + E + E + 1;
  if (need != n_elems) return fail(h, DM_ERR_INVALID, "dm_load_weights_din: n_elems does not match the DIN layout for (E, num_index)");
  free_weights(h);
  ALLOC(h, h->d_compact, (size_t)n_elems * sizeof(T));
  HIPCHK(h, hipMemcpy(h->d_compact, w, (size_t)n_elems * sizeof(T), hipMemcpyHostToDevice));
  // f32 table for the beam kernels
  if (sizeof(T) == 4) { h->d_emb32 = (float *)h->d_compact; h->emb32_owned = false; }
  else {
    ALLOC(h, h->d_emb32, (size_t)num_index * E * 4);
    h->emb32_owned = true;
    hipLaunchKernelGGL(dm_f64_to_f32_kernel, dim3(2048), dim3(256), 0, h->stream, (const double *)h->d_compact,
                       h->d_emb32, num_index * E);
    HIPCHK(h, hipGetLastError());
  }
  int rc = upload_derived<T>(h, E, w + num_index * E);
  if (rc != DM_OK) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->embed = E; h->embed_log = E; h->num_index = num_index; h->w_loaded = true; h->split_dirty = true; h->table_dense_change = true;
  return DM_OK;
}
template <typename T>
static int load_weights_any_t(dm_ctx *h, int E, int64_t num_index, const T *w, int64_t n_elems) {
  const int Ep = native_embed(E);
  if (Ep == E) return load_weights_t<T>(h, E, num_index, w, n_elems);
  if (n_elems != compact_len_for(num_index, E)) return fail(h, DM_ERR_INVALID, "dm_load_weights_din: n_elems does not match the DIN layout for (E, num_index)");
  std::vector<T> padded((size_t)compact_len_for(num_index, Ep));
  pad_compact<T>(w, E, Ep, num_index, padded.data());
  const int rc = load_weights_t<T>(h, Ep, num_index, padded.data(), (int64_t)padded.size());
  if (rc == DM_OK) h->embed_log = E;
  return rc;
}

int dm_load_weights_din(dm_handle_t h, int dtype, int E, int64_t num_index, const void *compact, int64_t n_elems) {
  if (!h) return DM_ERR_INVALID;
  DM_OWNER_ONLY(h, "dm_load_weights_din");
  if (!compact || num_index <= 0) return fail(h, DM_ERR_INVALID, "dm_load_weights_din: bad arguments");
  if (E < 1 || E > 128)
    return fail(h, DM_ERR_UNSUPPORTED, "dm_load_weights_din: embed size must be 1 .. 128 (sizes other than 16 / 32 / 64 / 128 are zero-padded to the next of them; at E = 256 the "
                "fp16 hi / lo planes of W1a are 256 KB against 64 KB of AccVGPRs per wave and 160 KB of LDS per CU: the weights would have to stream "
                "per 128-column slab, ~4.3 x the E = 128 time per scored row — DESIGN.md, not built)");
  if (dtype != DM_F32 && dtype != DM_F64) return fail(h, DM_ERR_INVALID, "dm_load_weights_din: dtype");
  HIPCHK(h, hipSetDevice(h->device));
  h->dtype = dtype;
  return dtype == DM_F32 ? load_weights_any_t<float>(h, E, num_index, (const float *)compact, n_elems)
                         : load_weights_any_t<double>(h, E, num_index, (const double *)compact, n_elems);
}

int dm_load_weights_din_dev(dm_handle_t h, int E, int64_t num_index, float *d_compact, int64_t n_elems) {
  if (!h) return DM_ERR_INVALID;
  DM_OWNER_ONLY(h, "dm_load_weights_din_dev");
  if (!d_compact || num_index <= 0) return fail(h, DM_ERR_INVALID, "dm_load_weights_din_dev: bad arguments");
  if (E != 16 && E != 32 && E != 64 && E != 128) return fail(h, DM_ERR_UNSUPPORTED, "dm_load_weights_din_dev: embed size must be 16, 32, 64 or 128");
  const int64_t need = num_index * E + (int64_t)E * E + (int64_t)E * 2 * E + E + E + 1;
  if (need != n_elems) return fail(h, DM_ERR_INVALID, "dm_load_weights_din_dev: n_elems does not match the DIN layout for (E, num_index)");
  HIPCHK(h, hipSetDevice(h->device));
  free_weights(h);
  h->dtype = DM_F32;
  h->d_compact = d_compact; h->d_emb32 = d_compact; h->emb32_owned = false;
  const int64_t tail = n_elems - num_index * E;
  std::vector<float> t((size_t)tail);
  HIPCHK(h, hipMemcpy(t.data(), d_compact + num_index * E, (size_t)tail * 4, hipMemcpyDeviceToHost));
  int rc = upload_derived<float>(h, E, t.data());
  if (rc != DM_OK) return rc;
  h->embed = E; h->embed_log = E; h->num_index = num_index; h->w_loaded = true; h->split_dirty = true; h->table_dense_change = true;
  return DM_OK;
}

// the fp64 counterpart (the reference's OTM model is DIN[Double]): same ownership rule; the f32 copy of the table that the
// throughput-mode beam kernels read is made here
int dm_load_weights_din_dev_f64(dm_handle_t h, int E, int64_t num_index, double *d_compact, int64_t n_elems) {
  if (!h) return DM_ERR_INVALID;
  DM_OWNER_ONLY(h, "dm_load_weights_din_dev_f64");
  if (!d_compact || num_index <= 0) return fail(h, DM_ERR_INVALID, "dm_load_weights_din_dev_f64: bad arguments");
  if (E != 16 && E != 32 && E != 64 && E != 128) return fail(h, DM_ERR_UNSUPPORTED, "dm_load_weights_din_dev_f64: embed size must be 16, 32, 64 or 128");
  const int64_t need = num_index * E + (int64_t)E * E + (int64_t)E * 2 * E + E + E + 1;
  if (need != n_elems) return fail(h, DM_ERR_INVALID, "dm_load_weights_din_dev_f64: n_elems does not match the DIN layout for (E, num_index)");
  HIPCHK(h, hipSetDevice(h->device));
  free_weights(h);
  h->dtype = DM_F64;
  h->d_compact = d_compact;
  ALLOC(h, h->d_emb32, (size_t)num_index * E * 4);
  h->emb32_owned = true;
  hipLaunchKernelGGL(dm_f64_to_f32_kernel, dim3(4096), dim3(256), 0, h->stream, (const double *)d_compact, h->d_emb32, num_index * E);
  HIPCHK(h, hipGetLastError());
  const int64_t tail = n_elems - num_index * E;
  std::vector<double> t((size_t)tail);
  HIPCHK(h, hipMemcpy(t.data(), d_compact + num_index * E, (size_t)tail * 8, hipMemcpyDeviceToHost));
  int rc = upload_derived<double>(h, E, t.data());
  if (rc != DM_OK) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->embed = E; h->embed_log = E; h->num_index = num_index; h->w_loaded = true; h->split_dirty = true; h->table_dense_change = true;
  return DM_OK;
}

template <typename T>
__global__ void dm_fill_normal_kernel(T *out, int64_t n, float mean, float std, unsigned long long seed) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
  for (; i < n; i += stride) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i / 2 + 1);   // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u1 = ((float)(unsigned)(z >> 40) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)(unsigned)((z >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.2831853071795864f * u2, &sn, &cs);
    out[i] = (T)(mean + std * r * cs);            // the f64 fill holds the f32 values, widened
    if (i + 1 < n) out[i + 1] = (T)(mean + std * r * sn);
  }
}

__global__ void dm_fill_tree_level_kernel(float *emb, int E, int64_t first, int64_t count, float rho, float sd, unsigned long long seed) {
  // one thread per PAIR of floats of the level's rows; row c = rho * row(parent) + sd * noise
  const int64_t n2 = count * E / 2;
  const float cn = sqrtf(1.0f - rho * rho);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = first + (2 * t) / E;
    const int e = (int)((2 * t) % E);
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(row * (E / 2) + e / 2 + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u1 = ((float)(unsigned)(z >> 40) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(unsigned)((z >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.2831853071795864f * u2, &sn, &cs);
    float p0 = 0.f, p1 = 0.f, c = 1.0f;
    if (row > 0) { const int64_t par = (row - 1) >> 1; p0 = emb[par * E + e]; p1 = emb[par * E + e + 1]; c = cn; }
    emb[row * E + e] = rho * p0 + c * sd * r * cs;
    emb[row * E + e + 1] = rho * p1 + c * sd * r * sn;
  }
}

int dm_fill_tree_normal(dm_handle_t h, float *d_emb, int E, int depth, float rho, float std, uint64_t seed) {
  if (!h) return DM_ERR_INVALID;
  if (!d_emb || E <= 0 || (E & 1) || depth < 0 || depth > 30 || rho < 0.f || rho >= 1.f) return fail(h, DM_ERR_INVALID, "dm_fill_tree_normal: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  for (int l = 0; l <= depth; l++) {
    const int64_t first = ((int64_t)1 << l) - 1, count = (int64_t)1 << l;
    int64_t blocks = (count * E / 2 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dm_fill_tree_level_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, d_emb, E, first, count, rho, std, (unsigned long long)seed);
    HIPCHK(h, hipGetLastError());
  }
  return DM_OK;
}

int dm_fill_normal(dm_handle_t h, float *d_ptr, int64_t n, float mean, float std, uint64_t seed) {
  if (!h) return DM_ERR_INVALID;
  if (!d_ptr || n < 0) return fail(h, DM_ERR_INVALID, "dm_fill_normal: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  if (n == 0) return DM_OK;
  hipLaunchKernelGGL(dm_fill_normal_kernel<float>, dim3(4096), dim3(256), 0, h->stream, d_ptr, n, mean, std, (unsigned long long)seed);
  HIPCHK(h, hipGetLastError());
  return DM_OK;
}

int dm_fill_normal_f64(dm_handle_t h, double *d_ptr, int64_t n, float mean, float std, uint64_t seed) {
  if (!h) return DM_ERR_INVALID;
  if (!d_ptr || n < 0) return fail(h, DM_ERR_INVALID, "dm_fill_normal_f64: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  if (n == 0) return DM_OK;
  hipLaunchKernelGGL(dm_fill_normal_kernel<double>, dim3(4096), dim3(256), 0, h->stream, d_ptr, n, mean, std, (unsigned long long)seed);
  HIPCHK(h, hipGetLastError());
  return DM_OK;
}

static bool fwd64_ready(dm_ctx *h, int L);                 // train_host.hip.inc
static int rows_fwd64(dm_ctx *h, const int32_t *d_codes, const int32_t *d_seqs, const unsigned *d_rowmask, int64_t B, int L, double *d_out);
template <typename T>
static int din_forward_t(dm_ctx *h, const int32_t *d_codes, const int32_t *d_seqs, const unsigned *d_rowmask, int64_t B,
                         int L, T *d_out) {
  if constexpr (sizeof(T) == 8) {
    // f64 models, batches of rows: the matrix-pipe forward (train_host.hip.inc: rows_fwd64 — the training kernel's forward half on
    // v_mfma_f64_16x16x4_f64) instead of the one-wave-per-row kernel below, which stays for single rows and for clones
    // (round 6: 34 M rows/s -> 10x; dm_din_forward, OTM child weights, evaluators).  DM_FWD64_SCALAR=1 keeps the scalar kernel.
    static const bool scalar_only = [] { const char *e_ = getenv("DM_FWD64_SCALAR"); return e_ && e_[0] == '1'; }();
    if (!scalar_only && B >= 256 && fwd64_ready(h, L)) return rows_fwd64(h, d_codes, d_seqs, d_rowmask, B, L, (double *)d_out);
  }
  DinFwdParams<T> p;
  const T *base = (const T *)h->d_compact;
  const int E = h->embed;
  p.emb = base; p.att_wT = (const T *)h->d_att_wT_t; p.l1T = (const T *)h->d_l1T_t;
  const T *l1_b = base + h->num_index * E + (int64_t)E * E + (int64_t)E * 2 * E;
  p.b1 = l1_b; p.w2 = l1_b + E;
  T b2;
  HIPCHK(h, hipMemcpy(&b2, l1_b + 2 * E, sizeof(T), hipMemcpyDeviceToHost));
  p.b2 = b2; p.E = E; p.L = L; p.B = B; p.codes = d_codes; p.seqs = d_seqs; p.rowmask = d_rowmask; p.out = d_out;
  p.sm_scale = (T)sm_scale64(h);
  int64_t blocks = (B + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  size_t lds = (size_t)4 * (3 * E + 32) * sizeof(T);
  hipLaunchKernelGGL(dm_din_forward_kernel<T>, dim3((unsigned)blocks), dim3(256), lds, h->stream, p);
  HIPCHK(h, hipGetLastError());
  return DM_OK;
}

template <int E>
static int launch_rows_E(dm_ctx *h, const RowsParams &p) {
  const int lds = 2 * E * E * 4;
  HIPCHK(h, hipFuncSetAttribute((const void *)dm_din_rows_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int64_t tiles = (p.B + 15) / 16;
  int64_t blocks = (tiles + DM_NWAVES - 1) / DM_NWAVES;
  if (blocks > h->n_cu) blocks = h->n_cu;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dm_din_rows_kernel<E>, dim3((unsigned)blocks), dim3(DM_BLOCK), lds, h->stream, p);
  HIPCHK(h, hipGetLastError());
  return DM_OK;
}

static bool use_split(const dm_ctx *h);
static int ensure_split_scales(dm_ctx *h);
static int split_shift(unsigned maxbits);
static bool weights_in_motion(const dm_ctx *h);
// fp16 planes of the general-rows split kernel (rows_kernel.hip.inc): follow the weights like the beam kernels' planes
static int ensure_rows_split(dm_ctx *h) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  int rc = ensure_split_scales(h);     // sh_e = the table's scale: one read of the table per weight change, shared with the beam kernels
  if (rc != DM_OK) return rc;
  if (!h->rows_split_dirty && h->d_rows_split) return DM_OK;
  const int E = h->embed;
  const size_t plane_bytes = (size_t)4 * E * E * 2;
  if (!h->d_rows_split) ALLOC(h, h->d_rows_split, plane_bytes + (size_t)E * E * 4);
  float *Mbuf = (float *)((char *)h->d_rows_split + plane_bytes);
  HIPCHK(h, hipMemsetAsync(h->d_maxabs, 0, 8, h->stream));
  hipLaunchKernelGGL(dm_rows_m_kernel, dim3(64), dim3(256), 0, h->stream, h->d_attA, h->d_w1aA, h->d_w1bA, E, Mbuf, h->d_maxabs);
  HIPCHK(h, hipGetLastError());
  unsigned mb[2];
  HIPCHK(h, hipMemcpyAsync(mb, h->d_maxabs, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  model_changed(h);
  h->sh_r = split_shift(mb[0] > mb[1] ? mb[0] : mb[1]);
  hipLaunchKernelGGL(dm_build_rows_planes_kernel, dim3(64), dim3(256), 0, h->stream, h->d_w1aA, (const float *)Mbuf, E, ldexpf(1.0f, h->sh_r),
                     (_Float16 *)h->d_rows_split);
  HIPCHK(h, hipGetLastError());
  h->rows_split_dirty = false;
  return DM_OK;
}

static int next_events(dm_ctx *h, hipEvent_t *a, hipEvent_t *b);
// HIP-event pair of kind 30 around a general-rows launch (dm_kernel_timing_get_kind: the roofline of JTM's scorer in bench.py)
struct RowsTimer {
  dm_ctx *h; hipEvent_t e0 = nullptr, e1 = nullptr; int rc = DM_OK;
  explicit RowsTimer(dm_ctx *h_) : h(h_) {
    const int k = h->ev_next_kind; h->ev_next_kind = 30; rc = next_events(h, &e0, &e1); h->ev_next_kind = k;
    if (rc == DM_OK && hipEventRecord(e0, h->stream) != hipSuccess) rc = fail(h, DM_ERR_HIP, "hipEventRecord failed");
  }
  int stop() { return (rc == DM_OK && hipEventRecord(e1, h->stream) != hipSuccess) ? fail(h, DM_ERR_HIP, "hipEventRecord failed") : rc; }
};

template <int E, int LC>
static int launch_rows_split_EL(dm_ctx *h, const RowsSplitParams &p, int64_t blocks) {
  const int lds = 4 * E * E * 2 + DM_NWAVES * 2 * (DM_MAXL + 2) * 16 * 4 + 2 * E * 4;      // weight planes + the per-wave index staging + b1, w2
  HIPCHK(h, hipFuncSetAttribute((const void *)dm_din_rows_split_l_kernel<E, LC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  RowsTimer tm(h);
  if (tm.rc != DM_OK) return tm.rc;
  hipLaunchKernelGGL((dm_din_rows_split_l_kernel<E, LC>), dim3((unsigned)blocks), dim3(DM_BLOCK), lds, h->stream, p);
  HIPCHK(h, hipGetLastError());
  return tm.stop();
}

// History lengths with a counted-load instance (rows_kernel.hip.inc: dm_din_rows_split_l_kernel); every other length takes the generic
// kernel — the same arithmetic in the same order, bit-identical results (tests/test_gpu_precision.py).  DM_ROWS_GENERIC=1 forces it.
template <int E>
static int launch_rows_split_E(dm_ctx *h, const RowsSplitParams &p) {
  int64_t tiles = (p.B + 15) / 16;
  int64_t blocks = (tiles + DM_NWAVES - 1) / DM_NWAVES;
  if (blocks > h->n_cu) blocks = h->n_cu;
  if (blocks < 1) blocks = 1;
  static const bool generic_only = [] { const char *e_ = getenv("DM_ROWS_GENERIC"); return e_ && e_[0] == '1'; }();
  if (!generic_only) {
    switch (p.L) {
      case 8: return launch_rows_split_EL<E, 8>(h, p, blocks);
      case 10: return launch_rows_split_EL<E, 10>(h, p, blocks);
      case 16: return launch_rows_split_EL<E, 16>(h, p, blocks);
      default: break;
    }
  }
  const int lds = 4 * E * E * 2;
  HIPCHK(h, hipFuncSetAttribute((const void *)dm_din_rows_split_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  RowsTimer tm(h);
  if (tm.rc != DM_OK) return tm.rc;
  hipLaunchKernelGGL(dm_din_rows_split_kernel<E>, dim3((unsigned)blocks), dim3(DM_BLOCK), lds, h->stream, p);
  HIPCHK(h, hipGetLastError());
  return tm.stop();
}

// f32 general-rows forward on device buffers (asynchronous on the handle's stream)
static int din_rows_dev(dm_ctx *h, const int32_t *d_codes, const int32_t *d_seqs, const unsigned *d_rowmask, int64_t B,
                        int L, float *d_out, int seq_div = 1) {
  // the default arithmetic for E = 32 / 64 / 128 (DM_SCORER_F32 keeps the fp32-input kernel below).  While a training loop keeps
  // moving the weights (AUTO mode) a batch below ~4 M rows is cheaper on the fp32-input kernel than the table scan for the new scale.
  if (use_split(h) && !(weights_in_motion(h) && B < ((int64_t)1 << 22))) {
    int rc = ensure_rows_split(h);
    if (rc != DM_OK) return rc;
    RowsSplitParams q;
    q.emb = h->d_emb32; q.planes = (const dm_h8 *)h->d_rows_split; q.b1 = h->d_b1; q.w2 = h->d_w2; q.b2 = h->b2;
    q.emb_scale = ldexpf(1.0f, h->sh_e); q.out_unscale = ldexpf(1.0f, -(h->sh_e + h->sh_r));
    q.num_index = h->num_index; q.codes = d_codes; q.seqs = d_seqs; q.rowmask = d_rowmask; q.B = B; q.L = L; q.out = d_out;
    q.sm_scale = sm_scale32(h); q.seq_div = seq_div;
    switch (h->embed) {
      DM_IF_ALL_E(case 32: return launch_rows_split_E<32>(h, q);)
      DM_IF_ALL_E(case 64: return launch_rows_split_E<64>(h, q);)
      case 128: return launch_rows_split_E<128>(h, q);
    }
  }
  RowsParams p;
  p.emb = h->d_emb32; p.attA = h->d_attA; p.w1aA = h->d_w1aA; p.w1bA = h->d_w1bA; p.b1 = h->d_b1; p.w2 = h->d_w2;
  p.b2 = h->b2; p.num_index = h->num_index; p.codes = d_codes; p.seqs = d_seqs; p.rowmask = d_rowmask; p.B = B; p.L = L;
  p.out = d_out; p.sm_scale = sm_scale32(h); p.seq_div = seq_div;
  switch (h->embed) {
    DM_IF_ALL_E(case 16: return launch_rows_E<16>(h, p);)
    DM_IF_ALL_E(case 32: return launch_rows_E<32>(h, p);)
    DM_IF_ALL_E(case 64: return launch_rows_E<64>(h, p);)
    case 128: return launch_rows_E<128>(h, p);
  }
  return fail(h, DM_ERR_UNSUPPORTED, "unsupported embed size");
}

int dm_din_forward(dm_handle_t h, const int32_t *codes, const int32_t *seqs, const int32_t *pad_flat_idx,
                   int64_t n_pad, int64_t B, int L, void *logits) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  if (!h->w_loaded) return fail(h, DM_ERR_STATE, "dm_din_forward: weights not loaded");
  if (!codes || !seqs || !logits || B < 0 || L <= 0 || L > 32 || n_pad < 0 || (n_pad > 0 && !pad_flat_idx))
    return fail(h, DM_ERR_INVALID, "dm_din_forward: bad arguments (L must be 1..32)");
  if (B == 0) return DM_OK;
  // LookupTable.embeddingLookup validates every index first (LookupTable.scala:29-53)
  for (int64_t i = 0; i < B; i++)
    if (codes[i] != -1 && (codes[i] < 0 || codes[i] >= h->num_index)) {
      char b[160]; snprintf(b, sizeof b, "embeddingLookup failed, valid index range is [0, %lld), row %lld got %d", (long long)h->num_index, (long long)i, codes[i]);
      return fail(h, DM_ERR_INDEX, b);
    }
  for (int64_t i = 0; i < B * L; i++)
    if (seqs[i] != -1 && (seqs[i] < 0 || seqs[i] >= h->num_index)) {
      char b[160]; snprintf(b, sizeof b, "embeddingLookup failed, valid index range is [0, %lld), row %lld got %d", (long long)h->num_index, (long long)(i / L), seqs[i]);
      return fail(h, DM_ERR_INDEX, b);
    }
  for (int64_t i = 0; i < n_pad; i++)
    if (pad_flat_idx[i] < 0 || pad_flat_idx[i] >= B * L) return fail(h, DM_ERR_INDEX, "dm_din_forward: mask index outside [0, B*L)");
  HIPCHK(h, hipSetDevice(h->device));
  int32_t *d_codes = nullptr, *d_seqs = nullptr, *d_pad = nullptr;
  unsigned *d_mask = nullptr;
  void *d_out = nullptr;
  const size_t esz = h->dtype == DM_F32 ? 4 : 8;
  int rc = DM_OK;
  do {
    if ((rc = dm_alloc(h, (void **)&d_codes, B * 4)) != DM_OK) break;
    if ((rc = dm_alloc(h, (void **)&d_seqs, B * L * 4)) != DM_OK) break;
    if ((rc = dm_alloc(h, (void **)&d_mask, B * 4)) != DM_OK) break;
    if ((rc = dm_alloc(h, &d_out, B * esz)) != DM_OK) break;
    if (hipMemcpyAsync(d_codes, codes, B * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(d_seqs, seqs, B * L * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemsetAsync(d_mask, 0, B * 4, h->stream) != hipSuccess) { rc = fail(h, DM_ERR_HIP, "dm_din_forward: upload failed"); break; }
    if (n_pad > 0) {
      if (dm_alloc(h, (void **)&d_pad, n_pad * 4) != DM_OK) { rc = DM_ERR_HIP; break; }
      if (hipMemcpyAsync(d_pad, pad_flat_idx, n_pad * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(h, DM_ERR_HIP, "upload failed"); break; }
      hipLaunchKernelGGL(dm_pad_rowmask_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, h->stream, d_pad, n_pad, L, d_mask);
    }
    rc = h->dtype == DM_F32 ? (L <= DM_MAXL ? din_rows_dev(h, d_codes, d_seqs, d_mask, B, L, (float *)d_out)
                                            : din_forward_t<float>(h, d_codes, d_seqs, d_mask, B, L, (float *)d_out))
                            : din_forward_t<double>(h, d_codes, d_seqs, d_mask, B, L, (double *)d_out);
    if (rc != DM_OK) break;
    if (hipMemcpyAsync(logits, d_out, B * esz, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(h, DM_ERR_HIP, std::string("dm_din_forward: ") + hipGetErrorString(hipGetLastError())); break; }
  } while (0);
  dm_free_ptr(d_codes); dm_free_ptr(d_seqs); dm_free_ptr(d_mask); dm_free_ptr(d_out); dm_free_ptr(d_pad);
  return rc;
}

// ------------------------------------------------------------ beam search
struct SearchPlan {
  int nteams, cap, pcap, grid, ws_cap, lds;
  bool wkernel;        // the one-wave-per-SIMD kernel with W1a in the AccVGPRs (beam_kernel_w.hip.inc)
};

// the scorer arithmetic the beam kernels will use for this handle's model (dm_set_scorer_mode)
static bool use_split(const dm_ctx *h) {
  return (h->scorer_mode == DM_SCORER_SPLIT_F16 || h->scorer_mode == DM_SCORER_AUTO) && h->embed % 32 == 0;
}

// AUTO mode inside a training loop: the split scorer's scales / fp16 copies are stale after every Adam step, and refreshing them is a
// pass (or three) over the whole table.  A request that is small next to that takes the fp32-input kernels, which read the fp32
// table as it is; both arithmetics meet the same tolerance.  An explicit DM_SCORER_SPLIT_F16 is always honoured.
static bool weights_in_motion(const dm_ctx *h) {
  return h->scorer_mode == DM_SCORER_AUTO && h->train_ready && (h->split_dirty || h->f32_mirror_dirty);
}
static bool split_for_call(const dm_ctx *h, int64_t U, int max_beam) {
  if (!use_split(h)) return false;
  if (weights_in_motion(h) || (h->scorer_mode == DM_SCORER_AUTO && h->train_ready && h->emb_split_dirty)) {
    const bool patchable = h->sh_e_valid && !h->table_dense_change && h->dtype == DM_F32;                       // only the Adam step's active rows are stale
    const double rows_ = patchable ? (double)h->active_rows_host : (double)h->num_index;
    const double rebuild_s = 3.0e-5 + 3.0 * rows_ * h->embed * 4 / (patchable ? 1.0e12 : 3.0e12);   // launches + read-back, then scan + read + write
    const double extra_s = (double)U * max_beam * 7.0e-9;                             // fp32-input kernel: ~7 ns more per (user, beam slot)
    if (extra_s < rebuild_s) return false;
  }
  return true;
}

// fp64 parity mode of the OTM search (otm64.hip.inc): in effect when f64 weights are loaded and the scorer mode is AUTO or F64
static bool use_f64_beam(const dm_ctx *h) {
  return h->dtype == DM_F64 && (h->scorer_mode == DM_SCORER_AUTO || h->scorer_mode == DM_SCORER_F64);
}

// Histories of 17 .. 32 positions run inside the fused LDS-fed kernel (two key tiles, beam_kernel.hip.inc) whenever its frontier fits
// LDS beside the second key tile; the per-level pipelines (tdm_pipeline.hip.inc, otm64.hip.inc) remain for beams beyond that and for
// A/B runs (DM_LONG_PIPELINE=1).
static bool long_history_pipeline(const dm_ctx *h, int max_beam, int L) {
  if (L <= DM_MAXL) return false;
  const char *e_ = getenv("DM_LONG_PIPELINE");          // read per call: the tests run both routes in one process
  if (e_ && e_[0] == '1') return true;
  int cap = ((2 * max_beam + 15) / 16) * 16;
  if (cap < 32) cap = 32;
  int pcap = 16;
  while (pcap < cap) pcap <<= 1;
  // (sized for the split scorer's layout; the fp32-input layout of the same request is no larger)
  return dm_beam_lds(h->embed, 1, cap, pcap, 4, true, 2).total > 160 * 1024;
}

static int plan_search(dm_ctx *h, int max_beam, int64_t U, int L, int n_levels, bool tdm, SearchPlan *pl) {
  int cap = ((2 * max_beam + 15) / 16) * 16;
  if (cap < 32) cap = 32;
  int pcap = 16;
  while (pcap < cap) pcap <<= 1;
  const int kt = L > DM_MAXL ? 2 : 1;          // histories of 17 .. 32 positions: the LDS-fed kernel's two-key-tile instance
  const int kq = kt > 1 ? 4 : (L + 3) / 4;
  int nteams = 0;
  pl->wkernel = false;
  h->call_split = split_for_call(h, U, max_beam);
  if (h->call_split && h->beam_w && kt == 1) {
    // split-fp16 scorer: one-wave teams, four per workgroup, W1a in registers; falls back when the frontier outgrows LDS
    BeamWLds l = dm_beamw_lds(h->embed, cap, pcap, kq);
    if (l.total <= 160 * 1024) { nteams = DMW_NWAVES; pl->lds = l.total; pl->wkernel = true; }
  }
  // LDS-fed kernel: teams of 8 / nteams waves.  A frontier of at most 256 slots fits ONE wave's register sort, so small beams
  // (the reference's serving default is candidateNum 20) run as eight one-wave teams: no team barriers at all on the latency
  // chain sort -> expand -> gather -> score of a level, and eight users per CU in flight instead of four
  if (!nteams)
  for (int cand = (pcap <= 256 ? 8 : 4); cand >= 1; cand >>= 1) {
    BeamLds l = dm_beam_lds(h->embed, cand, cap, pcap, kq, h->call_split, kt);
    if (l.total <= 160 * 1024) { nteams = cand; pl->lds = l.total; break; }
  }
  if (!nteams) return fail(h, DM_ERR_UNSUPPORTED, "beam too large for the LDS frontier (about 2*beam*28 bytes + weights must fit 160 KiB)");
  int64_t groups = (U + nteams - 1) / nteams;
  int grid = (int)(groups < h->n_cu ? groups : h->n_cu);
  if (grid < 1) grid = 1;
  pl->nteams = nteams; pl->cap = cap; pl->pcap = pcap; pl->grid = grid;
  pl->ws_cap = tdm ? (h->leaves_at_max_only ? cap : cap * (n_levels + 1)) : 16;
  return DM_OK;
}

static int ensure_ws(dm_ctx *h, size_t bytes) {
  if (h->ws_bytes >= bytes) return DM_OK;
  dm_free_ptr(h->d_ws); h->d_ws = nullptr; h->ws_bytes = 0;
  ALLOC(h, h->d_ws, bytes);
  h->ws_bytes = bytes;
  return DM_OK;
}

static int next_events(dm_ctx *h, hipEvent_t *a, hipEvent_t *b) {
  if (h->ev_used >= 4096) h->ev_used = 0;      // a service that never reads the timings keeps a bounded pool (oldest pairs are reused)
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    h->ev_pool.push_back({e0, e1});
  }
  *a = h->ev_pool[h->ev_used].first; *b = h->ev_pool[h->ev_used].second;
  if (h->ev_kind.size() <= h->ev_used) h->ev_kind.resize(h->ev_used + 1);
  h->ev_kind[h->ev_used] = h->ev_next_kind;
  h->ev_used++;
  return DM_OK;
}

template <int E, int KQ, bool SPLIT, int KT = 1>
static int launch_beam_EK(dm_ctx *h, const BeamParams &p_in, const SearchPlan &pl) {
  BeamParams p = p_in;
  p.static_users = (p.mode != 2 && !p.user_list && p.U <= (int64_t)pl.grid * pl.nteams) ? 1 : 0;
  HIPCHK(h, hipFuncSetAttribute((const void *)dm_beam_kernel<E, KQ, SPLIT, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, pl.lds));
  if (h->ev_next_kind == 0) {
    if (KT == 1) snprintf(h->last_kernel, sizeof(h->last_kernel), "dm_beam_kernel<%d, %d, %s>", E, KQ, SPLIT ? "true" : "false");
    else snprintf(h->last_kernel, sizeof(h->last_kernel), "dm_beam_kernel<%d, %d, %s, %d>", E, KQ, SPLIT ? "true" : "false", KT);
  }
  if (h->ev_skip && !h->time_direct) {       // single-request path: the launch and nothing else
    hipLaunchKernelGGL((dm_beam_kernel<E, KQ, SPLIT, KT>), dim3(pl.grid), dim3(DM_BLOCK), pl.lds, h->stream, p);
    HIPCHK(h, hipGetLastError());
    return DM_OK;
  }
  hipEvent_t e0, e1;
  int rc = next_events(h, &e0, &e1);
  if (rc != DM_OK) return rc;
  HIPCHK(h, hipEventRecord(e0, h->stream));
  hipLaunchKernelGGL((dm_beam_kernel<E, KQ, SPLIT, KT>), dim3(pl.grid), dim3(DM_BLOCK), pl.lds, h->stream, p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(e1, h->stream));
  return DM_OK;
}

template <int E, int KQ>
static int launch_beam_w_EK(dm_ctx *h, const BeamParams &p, const SearchPlan &pl) {
  HIPCHK(h, hipFuncSetAttribute((const void *)dm_beam_w_kernel<E, KQ>, hipFuncAttributeMaxDynamicSharedMemorySize, pl.lds));
  snprintf(h->last_kernel, sizeof(h->last_kernel), "dm_beam_w_kernel<%d, %d>", E, KQ);
  hipEvent_t e0, e1;
  int rc = next_events(h, &e0, &e1);
  if (rc != DM_OK) return rc;
  HIPCHK(h, hipEventRecord(e0, h->stream));
  hipLaunchKernelGGL((dm_beam_w_kernel<E, KQ>), dim3(pl.grid), dim3(DMW_BLOCK), pl.lds, h->stream, p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(e1, h->stream));
  return DM_OK;
}
template <int E>
static int launch_beam_w_E(dm_ctx *h, const BeamParams &p, const SearchPlan &pl) {
  switch ((p.L + 3) / 4) {
    case 1: return launch_beam_w_EK<E, 1>(h, p, pl);
    case 2: return launch_beam_w_EK<E, 2>(h, p, pl);
    case 3: return launch_beam_w_EK<E, 3>(h, p, pl);
    default: return launch_beam_w_EK<E, 4>(h, p, pl);
  }
}

template <int E, bool SPLIT>
static int launch_beam_E(dm_ctx *h, const BeamParams &p, const SearchPlan &pl) {
  if (p.L > DM_MAXL) return launch_beam_EK<E, 4, SPLIT, 2>(h, p, pl);          // 17 .. 32 history positions: two key tiles
  switch ((p.L + 3) / 4) {
    case 1: return launch_beam_EK<E, 1, SPLIT>(h, p, pl);
    case 2: return launch_beam_EK<E, 2, SPLIT>(h, p, pl);
    case 3: return launch_beam_EK<E, 3, SPLIT>(h, p, pl);
    default: return launch_beam_EK<E, 4, SPLIT>(h, p, pl);
  }
}

// ---- split-fp16 scorer: scales and the fp16 planes of W1a -------------------------------------------------------
__global__ void dm_maxabs_kernel(const float *x, int64_t n, unsigned *out) {
  unsigned m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(x[i]) & 0x7fffffffu;      // |x| as an ordered integer (NaN / inf sort highest)
    m = b > m ? b : m;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// planes[p][s][nt][lane][i], lane = (g, m): W1a[16nt + m][32s + 16(i>>2) + 4g + (i&3)] * 2^sh_w split into fp16 hi (p=0) and
// lo (p=1); the column order is the one the beam kernel's gathered rows have inside a lane (two float4 per k-step)
__global__ void dm_build_wsplit_kernel(const float *wfrag, int E, float scale, _Float16 *planes) {
  const int NT = E / 16, NS = E / 32;
  const int n = NS * NT * 64 * 8;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int i = t & 7, lane = (t >> 3) & 63, nt = (t >> 9) % NT, s = (t >> 9) / NT;
    const int jc = 2 * s + (i >> 2);
    const float x = wfrag[(((size_t)jc * NT + nt) * 64 + lane) * 4 + (i & 3)] * scale;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    planes[t] = hi;
    planes[(size_t)n + t] = lo;
  }
}

// The table as the one-wave-per-SIMD kernel gathers it: every fp32 value x already split into hi = RNE16(x 2^s) and
// lo = RNE16(x 2^s - hi), laid out so that the lane group g of a tile finds, per row and k-step, its two MFMA B operands as 32
// contiguous bytes: out[row][s][g][0..7] = hi of columns 32s + 16(i>>2) + 4g + (i&3), out[row][s][g][8..15] = lo of the same.
// Same bytes per row as the fp32 table (E * 4); identical values to the split the LDS-fed kernel does per tile.
__global__ void dm_build_emb_split_kernel(const float *emb, int64_t num_index, int E, float scale, _Float16 *out) {
  const int64_t n8 = num_index * (int64_t)(E / 8);        // groups of 8 values = one (row, s, g)
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / (E / 8);
    const int sg = (int)(t % (E / 8)), s_ = sg >> 2, g = sg & 3;
    const float *src = emb + row * E + 32 * s_ + 4 * g;
    _Float16 *dst = out + row * (int64_t)(2 * E) + (int64_t)sg * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float x = src[16 * (i >> 2) + (i & 3)] * scale;
      const _Float16 hi = (_Float16)x;
      dst[i] = hi;
      dst[8 + i] = (_Float16)(x - (float)hi);
    }
  }
}

// the same two passes over a LIST of rows (the rows an Adam step can have moved: ensure_split_scales / ensure_split inside a training loop)
__global__ void dm_maxabs_rows_kernel(const float *emb, const int32_t *rows, int64_t n_rows, int E, unsigned *out) {
  unsigned m = 0;
  const int64_t n = n_rows * E;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(emb[(int64_t)rows[t / E] * E + (t % E)]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
__global__ void dm_build_emb_split_rows_kernel(const float *emb, const int32_t *rows, int64_t n_rows, int E, float scale, _Float16 *out) {
  const int64_t n8 = n_rows * (int64_t)(E / 8);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = rows[t / (E / 8)];
    const int sg = (int)(t % (E / 8)), s_ = sg >> 2, g = sg & 3;
    const float *src = emb + row * E + 32 * s_ + 4 * g;
    _Float16 *dst = out + row * (int64_t)(2 * E) + (int64_t)sg * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float x = src[16 * (i >> 2) + (i & 3)] * scale;
      const _Float16 hi = (_Float16)x;
      dst[i] = hi;
      dst[8 + i] = (_Float16)(x - (float)hi);
    }
  }
}

// power-of-two shift that puts max|x| into [2^13, 2^14): every scaled value and every rounding of it stays below the fp16
// maximum, and fp16 subnormals only start 2^27 below the largest element
static int split_shift(unsigned maxbits) {
  if (maxbits == 0 || maxbits >= 0x7f800000u) return 0;
  float m;
  memcpy(&m, &maxbits, 4);
  int e;
  frexpf(m, &e);               // m = f * 2^e, f in [0.5, 1)
  int sh = 14 - e;
  if (sh > 40) sh = 40;
  if (sh < -40) sh = -40;
  return sh;
}

// scales (2^sh_e from max|emb|: one read of the table; 2^sh_w from max|W1a|) and the fp16 planes of W1a: what every split kernel
// needs.  The pre-split copy of the table (ensure_split) is a second, larger step only the beam kernels take.
static int ensure_split_scales(dm_ctx *h) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  if (!h->split_dirty && h->d_wsplit) return DM_OK;
  model_changed(h);
  const int E = h->embed;
  if (E % 32 != 0) return fail(h, DM_ERR_UNSUPPORTED, "the split-fp16 scorer needs an embedding size that is a multiple of 32");
  if (!h->d_wsplit) ALLOC(h, h->d_wsplit, (size_t)E * E * 4);
  if (!h->d_maxabs) ALLOC(h, h->d_maxabs, 8);
  // Inside a training loop whose Adam steps visit the active rows only, nothing else of the table has moved since the last full
  // scan: the scale 2^sh_e stays (a power of two: exact) as long as the active rows still fit it, and only those rows are re-split.
  bool patch = h->train_ready && h->sh_e_valid && !h->table_dense_change && h->d_active_list && h->dtype == DM_F32;
  unsigned mb[2];
  for (;;) {
    HIPCHK(h, hipMemsetAsync(h->d_maxabs, 0, 8, h->stream));
    if (patch) {
      if (h->active_rows_host)
        hipLaunchKernelGGL(dm_maxabs_rows_kernel, dim3(1024), dim3(256), 0, h->stream, h->d_emb32, h->d_active_list, (int64_t)h->active_rows_host, E, h->d_maxabs);
    } else
      hipLaunchKernelGGL(dm_maxabs_kernel, dim3(4096), dim3(256), 0, h->stream, h->d_emb32, h->num_index * (int64_t)E, h->d_maxabs);
    hipLaunchKernelGGL(dm_maxabs_kernel, dim3(16), dim3(256), 0, h->stream, (const float *)h->d_wfrag, (int64_t)E * E, h->d_maxabs + 1);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(mb, h->d_maxabs, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (patch && mb[0] != 0 && split_shift(mb[0]) < h->sh_e) { patch = false; continue; }    // an active row outgrew the scale: full scan
    break;
  }
  if (!patch) {
    h->sh_e = split_shift(mb[0]);
    h->sh_e_valid = true; h->table_dense_change = false; h->emb_split_need_full = true;
  }
  h->emb_split_patch = patch;
  h->sh_w = split_shift(mb[1]);
  hipLaunchKernelGGL(dm_build_wsplit_kernel, dim3(64), dim3(256), 0, h->stream, (const float *)h->d_wfrag, E, ldexpf(1.0f, h->sh_w),
                     (_Float16 *)h->d_wsplit);
  HIPCHK(h, hipGetLastError());
  h->split_dirty = false;
  h->emb_split_dirty = true;
  h->rows_split_dirty = true;
  return DM_OK;
}

static int ensure_split(dm_ctx *h) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  int rc = ensure_split_scales(h);
  if (rc != DM_OK) return rc;
  if (!h->emb_split_dirty && h->d_emb_split) return DM_OK;
  model_changed(h);
  const int E = h->embed;
  // the beam kernels gather pre-split rows: a second copy of the table (same size), refreshed whenever the weights change — as a whole,
  // or in the active rows only when nothing else can have moved
  const size_t bytes = (size_t)h->num_index * E * 4;
  if (h->emb_split_bytes != bytes) {
    dm_free_ptr(h->d_emb_split); h->d_emb_split = nullptr; h->emb_split_bytes = 0;
    ALLOC(h, h->d_emb_split, bytes);
    h->emb_split_bytes = bytes;
    h->emb_split_need_full = true;
  }
  if (h->emb_split_need_full || !h->emb_split_patch) {
    hipLaunchKernelGGL(dm_build_emb_split_kernel, dim3(8192), dim3(256), 0, h->stream, h->d_emb32, h->num_index, E, ldexpf(1.0f, h->sh_e),
                       (_Float16 *)h->d_emb_split);
    h->emb_split_need_full = false;
  } else if (h->active_rows_host) {
    hipLaunchKernelGGL(dm_build_emb_split_rows_kernel, dim3(1024), dim3(256), 0, h->stream, h->d_emb32, h->d_active_list, (int64_t)h->active_rows_host,
                       E, ldexpf(1.0f, h->sh_e), (_Float16 *)h->d_emb_split);
  }
  HIPCHK(h, hipGetLastError());
  h->emb_split_dirty = false;
  return DM_OK;
}

static int ensure_f32_mirror(dm_ctx *h);
static int launch_beam(dm_ctx *h, BeamParams &p, const SearchPlan &pl) {
  // the host-polled epilogue (results, system-scope fence, count as the flag) exists for the TDM final selection only: the mode-1 (OTM)
  // epilogue stores its counts before the ids and without a fence
  if (p.host_direct && p.mode != 0) return fail(h, DM_ERR_STATE, "launch_beam: the host-mapped single-request path is a mode-0 (TDM) path");
  {   // f64 model trained since the f32 copies were made: refresh them (and the scorer parameters fill_common took from them)
    const bool was = h->f32_mirror_dirty;
    int rc_ = ensure_f32_mirror(h);
    if (rc_ != DM_OK) return rc_;
    if (was) p.b2 = h->b2;
  }
  // the brute-force recall oracle (mode 2) always scores with the fp32-input MFMA
  if (h->scorer_mode == DM_SCORER_SPLIT_F16 && h->embed % 32 != 0)
    return fail(h, DM_ERR_UNSUPPORTED, "the split-fp16 scorer needs an embedding size of 32, 64 or 128");
  if (h->call_split && p.mode != 2) {
    int rc = ensure_split(h);
    if (rc != DM_OK) return rc;
    p.wsplit = (const dm_h8 *)h->d_wsplit;
    p.emb_split = (const dm_h8 *)h->d_emb_split;
    p.emb_scale = ldexpf(1.0f, h->sh_e); p.score_unscale = ldexpf(1.0f, -2 * h->sh_e);
    p.acc_scale = ldexpf(1.0f, h->sh_e + h->sh_w); p.out_unscale = ldexpf(1.0f, -(h->sh_e + h->sh_w));
    if (pl.wkernel) {
      // users the W kernel cannot score in fp16 throughout are queued in [count | next | ids ...] and scored by the LDS-fed kernel
      const size_t need = 16 + (size_t)p.U * 4;
      if (h->defer_bytes < need) {
        dm_free_ptr(h->d_defer); h->d_defer = nullptr; h->defer_bytes = 0;
        ALLOC(h, h->d_defer, need + need / 4);
        h->defer_bytes = need + need / 4;
      }
      HIPCHK(h, hipMemsetAsync(h->d_defer, 0, 16, h->stream));
      p.defer_count = (unsigned long long *)h->d_defer;
      p.defer_users = (int32_t *)((char *)h->d_defer + 16);
      int rc = DM_OK;
      switch (h->embed) {
        DM_IF_ALL_E(case 32: rc = launch_beam_w_E<32>(h, p, pl); break;)
        DM_IF_ALL_E(case 64: rc = launch_beam_w_E<64>(h, p, pl); break;)
        default: rc = launch_beam_w_E<128>(h, p, pl); break;
      }
      if (rc != DM_OK) return rc;
      // second pass (an empty list costs one launch): same parameters, the list as the work queue, its own frontier layout
      SearchPlan pl2;
      {
        const int kq = (p.L + 3) / 4;
        int nteams = 0;
        for (int cand = 4; cand >= 1; cand >>= 1) {
          BeamLds l = dm_beam_lds(h->embed, cand, pl.cap, pl.pcap, kq, true);
          if (l.total <= 160 * 1024) { nteams = cand; pl2.lds = l.total; break; }
        }
        if (!nteams) return fail(h, DM_ERR_UNSUPPORTED, "beam too large for the LDS frontier");
        pl2 = pl; pl2.nteams = nteams; pl2.wkernel = false;
        BeamLds l = dm_beam_lds(h->embed, nteams, pl.cap, pl.pcap, kq, true);
        pl2.lds = l.total;
        // the workspace was sized for grid x 4 one-wave teams; the second pass may use at most as many (block, team) slots
        if (pl2.grid * pl2.nteams > pl.grid * pl.nteams) pl2.grid = pl.grid * pl.nteams / pl2.nteams;
        if (pl2.grid < 1) pl2.grid = 1;
      }
      BeamParams p2 = p;
      p2.nteams = pl2.nteams;
      p2.user_count = (const unsigned long long *)h->d_defer;
      p2.user_list = (const int32_t *)((char *)h->d_defer + 16);
      p2.next_user = (unsigned long long *)((char *)h->d_defer + 8);
      p2.defer_count = nullptr; p2.defer_users = nullptr;
      h->ev_next_kind = 1;
      switch (h->embed) {
        DM_IF_ALL_E(case 32: rc = launch_beam_E<32, true>(h, p2, pl2); break;)
        DM_IF_ALL_E(case 64: rc = launch_beam_E<64, true>(h, p2, pl2); break;)
        default: rc = launch_beam_E<128, true>(h, p2, pl2); break;
      }
      h->ev_next_kind = 0;
      return rc;
    }
    switch (h->embed) {
      DM_IF_ALL_E(case 32: return launch_beam_E<32, true>(h, p, pl);)
      DM_IF_ALL_E(case 64: return launch_beam_E<64, true>(h, p, pl);)
      case 128: return launch_beam_E<128, true>(h, p, pl);
    }
    return fail(h, DM_ERR_UNSUPPORTED, "the split-fp16 scorer needs an embedding size of 32, 64 or 128");
  }
  switch (h->embed) {
    DM_IF_ALL_E(case 16: return launch_beam_E<16, false>(h, p, pl);)
    DM_IF_ALL_E(case 32: return launch_beam_E<32, false>(h, p, pl);)
    DM_IF_ALL_E(case 64: return launch_beam_E<64, false>(h, p, pl);)
    case 128: return launch_beam_E<128, false>(h, p, pl);
  }
  return fail(h, DM_ERR_UNSUPPORTED, "unsupported embed size");
}

int dm_set_scorer_mode(dm_handle_t h, int mode) {
  if (!h) return DM_ERR_INVALID;
  if (mode != DM_SCORER_F32 && mode != DM_SCORER_SPLIT_F16 && mode != DM_SCORER_AUTO && mode != DM_SCORER_F64) return fail(h, DM_ERR_INVALID, "dm_set_scorer_mode: unknown mode");
  if (mode == DM_SCORER_F64 && h->w_loaded && h->dtype != DM_F64)
    return fail(h, DM_ERR_UNSUPPORTED, "dm_set_scorer_mode: the fp64 scorer needs f64 weights");
  if (mode == DM_SCORER_SPLIT_F16 && h->w_loaded && h->embed % 32 != 0)
    return fail(h, DM_ERR_UNSUPPORTED, "dm_set_scorer_mode: the split-fp16 scorer needs an embedding size of 32, 64 or 128");
  h->scorer_mode = mode;
  if (h->parent) h->seen_epoch = 0;      // a clone: its next call makes sure the parent holds the copies this setting reads
  return DM_OK;
}

int dm_get_scorer_mode(dm_handle_t h, int *mode, int *effective, int *shift_emb, int *shift_w) {
  if (!h || !mode) return DM_ERR_INVALID;
  *mode = h->scorer_mode;
  if (effective) *effective = (h->w_loaded && use_f64_beam(h)) ? DM_SCORER_F64 : (h->w_loaded && use_split(h)) ? DM_SCORER_SPLIT_F16 : DM_SCORER_F32;
  if (shift_emb) *shift_emb = h->sh_e;
  if (shift_w) *shift_w = h->sh_w;
  return DM_OK;
}

static void fill_common(dm_ctx *h, BeamParams &p) {
  memset(&p, 0, sizeof p);
  p.emb = h->d_emb32; p.wfrag = h->d_wfrag; p.afrag = h->d_afrag; p.bfrag = h->d_bfrag; p.b1 = h->d_b1; p.w2 = h->d_w2;
  p.b2 = h->b2; p.num_index = h->num_index;
  p.exists_bits = h->d_exists; p.leaf_bits = h->d_leaf; p.node_id = h->d_node_id; p.id_to_code = h->d_id_to_code;
  p.n_slots = h->n_slots; p.non_leaf_offset = h->non_leaf_offset; p.max_code = h->max_code; p.max_level = h->max_level;
  p.scored_rows = h->d_rows;
  p.sm_scale = sm_scale32(h);
  p.phase_cycles = h->d_phase;
  p.next_user = h->d_rows + 1;
}

static int tdm_pipeline_dev(dm_ctx *h, const int32_t *d_seq, int64_t U, int L, const dm_tdm_search_opts *o, int max_beam,
                            const int64_t *d_coff, const int32_t *d_cids, int32_t *d_ids, float *d_scores, int32_t *d_counts,
                            int trace_levels, int cap, int32_t *d_tc, float *d_ts, int32_t *d_tn);

static int tdm_search_dev(dm_ctx *h, const int32_t *d_seq, int64_t U, int L, const dm_tdm_search_opts *o, int max_beam,
                          const int64_t *d_coff, const int32_t *d_cids, int32_t *d_ids, float *d_scores, int32_t *d_counts,
                          int trace_levels, int32_t *d_tc, float *d_ts, int32_t *d_tn, bool *direct = nullptr) {
  if (!h->tree_loaded || !h->ids_loaded || !h->w_loaded) return fail(h, DM_ERR_STATE, "tdm beam search: tree, id maps and weights must be loaded first");
  if (U < 0 || L <= 0 || L > DM_PIPE_MAXL || !o || o->beam <= 0 || o->topk <= 0) return fail(h, DM_ERR_INVALID, "tdm beam search: bad arguments (L must be 1..32)");
  if (U == 0) return DM_OK;          // an empty batch is not an error
  if (h->n_slots > h->num_index) return fail(h, DM_ERR_INDEX, "tdm beam search: tree codes exceed the embedding table (embeddingLookup would fail)");
  if (h->max_code >= h->num_index) return fail(h, DM_ERR_INDEX, "tdm beam search: id map codes exceed the embedding table");
  if (long_history_pipeline(h, max_beam, L)) {
    // histories of 17 .. 32 positions whose frontier does not fit LDS beside two key tiles: the per-level pipeline (tdm_pipeline.hip.inc)
    if (direct) { *direct = false; return DM_OK; }          // (the caller takes the staged path and comes back)
    HIPCHK(h, hipMemsetAsync(d_ids, 0xFF, (size_t)U * o->topk * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(d_scores, 0, (size_t)U * o->topk * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(d_counts, 0, (size_t)U * 4, h->stream));
    const int cap_ = ((2 * max_beam + 15) / 16) * 16 < 32 ? 32 : ((2 * max_beam + 15) / 16) * 16;
    return tdm_pipeline_dev(h, d_seq, U, L, o, max_beam, d_coff, d_cids, d_ids, d_scores, d_counts, trace_levels, cap_, d_tc, d_ts, d_tn);
  }
  int start, level;
  level_start_int(o->beam, &start, &level);
  int n_levels = h->max_level - level + 1;
  if (n_levels < 0) n_levels = 0;
  SearchPlan pl;
  int rc = plan_search(h, max_beam, U, L, n_levels, true, &pl);
  if (rc != DM_OK) return rc;
  const size_t per = (size_t)pl.grid * pl.nteams * pl.ws_cap;
  rc = ensure_ws(h, per * 16);
  if (rc != DM_OK) return rc;
  BeamParams p;
  fill_common(h, p);
  p.seq = d_seq; p.U = U; p.L = L; p.use_mask = o->use_mask; p.beam = o->beam; p.topk = o->topk;
  p.widen = (o->widen_consumed && d_coff) ? 1 : 0;
  p.consumed_off = d_coff; p.consumed_ids = d_cids; p.mode = 0; p.nteams = pl.nteams; p.cap = pl.cap; p.pcap = pl.pcap; p.leaf_fast = h->leaves_at_max_only ? 1 : 0;
  p.out_ids = d_ids; p.out_scores = d_scores; p.out_counts = d_counts; p.out_stride = o->topk;
  p.ws_code = (int32_t *)h->d_ws; p.ws_score = (float *)h->d_ws + per; p.ws_khi = (uint32_t *)h->d_ws + 2 * per;
  p.ws_klo = (uint32_t *)h->d_ws + 3 * per; p.ws_cap = pl.ws_cap;
  p.trace_codes = d_tc; p.trace_scores = d_ts; p.trace_counts = d_tn; p.trace_levels = trace_levels;
  if (direct) {
    // single-request path (tdm_search_host): request and results live in host-mapped pinned memory, one team per user without the
    // work queue, the kernel fills every output slot and publishes the counts last — ONE launch, no memsets, no copies, no events
    *direct = *direct && !pl.wkernel && U <= (int64_t)pl.grid * pl.nteams && !d_coff && !d_tn;
    if (!*direct) return DM_OK;             // nothing launched: the caller takes the staged path
    {
      p.host_direct = 1;
      p.scored_rows = h->d_rows + 3;          // not zeroed on this path: keep dm_last_scored_rows' counter out of it
      h->ev_skip = true;
      const int rc_ = launch_beam(h, p, pl);
      h->ev_skip = false;
      return rc_;
    }
  }
  if (h->rows_keep) HIPCHK(h, hipMemsetAsync(h->d_rows + 1, 0, 8, h->stream));      // (the work-queue head only)
  else HIPCHK(h, hipMemsetAsync(h->d_rows, 0, 16, h->stream));
  HIPCHK(h, hipMemsetAsync(d_ids, 0xFF, (size_t)U * o->topk * 4, h->stream));
  HIPCHK(h, hipMemsetAsync(d_scores, 0, (size_t)U * o->topk * 4, h->stream));
  HIPCHK(h, hipMemsetAsync(d_counts, 0, (size_t)U * 4, h->stream));
  return launch_beam(h, p, pl);
}

// ---- pipelined host-buffer searches.  The reference-shaped entry points take and return host arrays; at serving batch sizes the results
// are the bulk of a call (131 072 users x 200 x 8 B = 210 MB) and used to come down AFTER the kernel, through pageable memory: 98.6 ms per
// call against 77.5 ms for the kernel alone (round-4 bench).  Large requests are now cut into chunks of users: every chunk's kernel is
// enqueued up front on the handle's stream with an event behind it, and the host thread downloads chunk k on a second, non-blocking
// stream as soon as its event fires — under the kernels of the chunks behind it.  Only the last chunk's download is exposed.
// DM_HOST_PIPELINE=0 keeps the one-launch path (A/B).
// Chunk k covers users [off[k], off[k + 1]): equal chunks of about 24 MB of results, then a SHORT last one (its download is the only
// one nothing hides; 8 192 users still fill the persistent grid eight users deep).  n = 1: the request is not cut.
static int host_pipe_plan(size_t bytes_per_user, int64_t U, int64_t *off /* [18] */) {
  static const bool disabled = [] { const char *e = getenv("DM_HOST_PIPELINE"); return e && e[0] == '0'; }();
  off[0] = 0; off[1] = U;
  const size_t out_bytes = bytes_per_user * (size_t)U;
  if (disabled || U < 16384 || out_bytes < (32u << 20)) return 1;
  const int64_t last = U >= 65536 ? 8192 : 4096;
  const int64_t rest = U - last;
  int n = (int)((bytes_per_user * (size_t)rest + (24u << 20) - 1) / (24u << 20));
  if (n > 16) n = 16;
  if ((int64_t)n > rest / 4096) n = (int)(rest / 4096);
  if (n < 1) n = 1;
  const int64_t Uc = (((rest + n - 1) / n) + 255) / 256 * 256;
  int k = 0;
  for (; k < n && (int64_t)k * Uc < rest; k++) off[k] = (int64_t)k * Uc;
  off[k] = rest; off[k + 1] = U;
  return k + 1;
}
static int host_pipe_ensure(dm_ctx *h, int n) {
  if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  while ((int)h->chunk_ev.size() < n) {
    hipEvent_t e;
    HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->chunk_ev.push_back(e);
  }
  return DM_OK;
}

static int host_max_beam(const dm_tdm_search_opts *o, const int64_t *coff, int64_t U) {
  int mb = o->beam;
  if (o->widen_consumed && coff)
    for (int64_t u = 0; u < U; u++) {
      int64_t w = ((coff[u + 1] - coff[u]) + o->topk) / 2;   // Recommender.scala:31
      if (w > mb) mb = (int)w;
    }
  return mb;
}

int dm_tdm_beam_search_dev(dm_handle_t h, const int32_t *d_seq_item_ids, int64_t U, int L, const dm_tdm_search_opts *opts,
                           const int64_t *d_consumed_off, const int32_t *d_consumed_ids, int32_t *d_out_item_ids,
                           float *d_out_scores, int32_t *d_out_counts) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  if (!d_seq_item_ids || !opts || !d_out_item_ids || !d_out_scores || !d_out_counts) return fail(h, DM_ERR_INVALID, "dm_tdm_beam_search_dev: NULL argument");
  HIPCHK(h, hipSetDevice(h->device));
  int mb = opts->beam;
  if (opts->widen_consumed && d_consumed_off) {
    std::vector<int64_t> coff((size_t)U + 1);
    HIPCHK(h, hipMemcpy(coff.data(), d_consumed_off, (U + 1) * 8, hipMemcpyDeviceToHost));
    mb = host_max_beam(opts, coff.data(), U);
  }
  return tdm_search_dev(h, d_seq_item_ids, U, L, opts, mb, d_consumed_off, d_consumed_ids, d_out_item_ids, d_out_scores,
                        d_out_counts, 0, nullptr, nullptr, nullptr);
}

static int tdm_search_host(dm_ctx *h, const int32_t *seq, int64_t U, int L, const dm_tdm_search_opts *opts,
                           const int64_t *coff, const int32_t *cids, int32_t *out_ids, float *out_scores,
                           int32_t *out_counts, int max_levels, int32_t *tc, float *ts, int32_t *tn) {
  if (U == 0 && L > 0 && opts && opts->beam > 0 && opts->topk > 0) return DM_OK;          // an empty batch is not an error
  if (!seq || !opts || !out_ids || !out_scores || !out_counts || U <= 0 || L <= 0) return fail(h, DM_ERR_INVALID, "dm_tdm_beam_search: bad arguments");
  if (opts->beam <= 0 || opts->topk <= 0) return fail(h, DM_ERR_INVALID, "dm_tdm_beam_search: beam and topk must be positive");
  if ((coff == nullptr) != (cids == nullptr) && coff && coff[U] > 0) return fail(h, DM_ERR_INVALID, "dm_tdm_beam_search: consumed_off / consumed_ids must both be given");
  HIPCHK(h, hipSetDevice(h->device));
  const int mb = host_max_beam(opts, coff, U);
  const int cap = ((2 * mb + 15) / 16) * 16 < 32 ? 32 : ((2 * mb + 15) / 16) * 16;
  int32_t *d_seq = nullptr, *d_cids = nullptr, *d_ids = nullptr, *d_counts = nullptr, *d_tc = nullptr, *d_tn = nullptr;
  int64_t *d_coff = nullptr;
  float *d_scores = nullptr, *d_ts = nullptr;
  int rc = DM_OK;
  const size_t nout = (size_t)U * opts->topk;
  do {
    // one arena for the request's device buffers, kept in the handle
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int64_t nc = coff ? coff[U] : 0;
    const size_t b_seq = al((size_t)U * L * 4), b_out = al(nout * 4), b_cnt = al((size_t)U * 4);
    const size_t b_coff = coff ? al((size_t)(U + 1) * 8) : 0, b_cids = coff ? al((size_t)(nc > 0 ? nc : 1) * 4) : 0;
    const size_t need = b_seq + 2 * b_out + b_cnt + b_coff + b_cids;
    if (h->req_bytes < need) {
      dm_free_ptr(h->d_req); h->d_req = nullptr; h->req_bytes = 0;
      if ((rc = dm_alloc(h, &h->d_req, need + need / 2)) != DM_OK) break;
      h->req_bytes = need + need / 2;
    }
    char *base = (char *)h->d_req;
    d_seq = (int32_t *)base; base += b_seq;
    d_ids = (int32_t *)base; base += b_out;
    d_scores = (float *)base; base += b_out;
    d_counts = (int32_t *)base; base += b_cnt;
    // small requests (the reference's one-user-per-call serving loop): the request goes up and the three result arrays come down
    // through ONE pinned staging block — a pageable copy costs 10-15 us each, and a single-user search is 60 us of kernel
    const size_t down = 2 * b_out + b_cnt;
    const bool staged = !coff && !tn && b_seq + down <= (256u << 10);
    if (staged && (rc = ensure_stage(h)) != DM_OK) break;
    if (staged) memcpy(h->h_stage, seq, (size_t)U * L * 4);
    if (staged && h->direct_ok && U <= 8) {
      // Single-request path (the reference's serving loop: one user per call, examples/.../tdm/package.scala:114-124).  The staging
      // block is host-mapped: the kernel reads the request from it and writes the results into it; the host polls the counts.
      int32_t *m_ids = (int32_t *)(h->h_stage + b_seq);
      float *m_sc = (float *)(h->h_stage + b_seq + b_out);
      volatile int32_t *m_cnt = (volatile int32_t *)(h->h_stage + b_seq + 2 * b_out);
      for (int64_t u = 0; u < U; u++) m_cnt[u] = -1;
      bool direct = true;
      rc = tdm_search_dev(h, (const int32_t *)h->d_stage, U, L, opts, mb, nullptr, nullptr, (int32_t *)(h->d_stage + b_seq),
                          (float *)(h->d_stage + b_seq + b_out), (int32_t *)(h->d_stage + b_seq + 2 * b_out), 0, nullptr, nullptr, nullptr, &direct);
      if (rc != DM_OK) break;
      if (direct) {
        bool done = false;
        for (long spin = 0; !done; spin++) {
          done = true;
          for (int64_t u = 0; u < U; u++) done = done && m_cnt[u] >= 0;
          if (!done && (spin & 0xFFF) == 0xFFF && hipStreamQuery(h->stream) != hipErrorNotReady) {
            // the stream is idle (or failed) and the flags never came: a kernel fault
            hipError_t e = hipStreamSynchronize(h->stream);
            done = true;
            for (int64_t u = 0; u < U; u++) done = done && m_cnt[u] >= 0;
            if (!done) { rc = fail(h, DM_ERR_HIP, std::string("tdm beam search (single-request path): ") + (e != hipSuccess ? hipGetErrorString(e) : "kernel finished without results")); break; }
          }
        }
        if (rc != DM_OK) break;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        memcpy(out_ids, m_ids, nout * 4); memcpy(out_scores, m_sc, nout * 4);
        for (int64_t u = 0; u < U; u++) out_counts[u] = m_cnt[u];
        break;
      }
      // the plan did not allow it (one-wave-per-SIMD kernel, too many users for the grid): fall through to the staged path
    }
    if (hipMemcpyAsync(d_seq, staged ? (const void *)h->h_stage : (const void *)seq, (size_t)U * L * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(h, DM_ERR_HIP, "upload failed"); break; }
    int64_t coff_[18];
    const int n_chunks = (staged || coff || tn || long_history_pipeline(h, mb, L)) ? 1 : host_pipe_plan((size_t)opts->topk * 8, U, coff_);
    if (n_chunks > 1) {
      if ((rc = host_pipe_ensure(h, n_chunks)) != DM_OK) break;
      int launched = 0;
      for (int k = 0; k < n_chunks && rc == DM_OK; k++) {
        const int64_t u0 = coff_[k], uk = coff_[k + 1] - u0;
        if (uk <= 0) break;
        h->rows_keep = k > 0;
        rc = tdm_search_dev(h, d_seq + u0 * L, uk, L, opts, mb, nullptr, nullptr, d_ids + u0 * opts->topk, d_scores + u0 * opts->topk, d_counts + u0,
                            0, nullptr, nullptr, nullptr);
        h->rows_keep = false;
        if (rc == DM_OK && hipEventRecord(h->chunk_ev[(size_t)k], h->stream) != hipSuccess) rc = fail(h, DM_ERR_HIP, "tdm beam search: event record failed");
        if (rc == DM_OK) launched++;
      }
      hipError_t e = hipSuccess;
      for (int k = 0; k < launched && e == hipSuccess; k++) {
        const int64_t u0 = coff_[k], uk = coff_[k + 1] - u0;
        e = hipStreamWaitEvent(h->copy_stream, h->chunk_ev[(size_t)k], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(out_ids + u0 * opts->topk, d_ids + u0 * opts->topk, (size_t)uk * opts->topk * 4, hipMemcpyDeviceToHost, h->copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_scores + u0 * opts->topk, d_scores + u0 * opts->topk, (size_t)uk * opts->topk * 4, hipMemcpyDeviceToHost, h->copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_counts + u0, d_counts + u0, (size_t)uk * 4, hipMemcpyDeviceToHost, h->copy_stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(h->copy_stream);
      const hipError_t e2 = hipStreamSynchronize(h->stream);
      if (rc == DM_OK && (e != hipSuccess || e2 != hipSuccess)) rc = fail(h, DM_ERR_HIP, std::string("tdm beam search (pipelined download): ") + hipGetErrorString(e != hipSuccess ? e : e2));
      break;
    }
    if (coff) {
      d_coff = (int64_t *)base; base += b_coff;
      d_cids = (int32_t *)base; base += b_cids;
      if (hipMemcpyAsync(d_coff, coff, (size_t)(U + 1) * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(h, DM_ERR_HIP, "upload failed"); break; }
      if (nc > 0 && hipMemcpyAsync(d_cids, cids, (size_t)nc * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail(h, DM_ERR_HIP, "upload failed"); break; }
    }
    if (tn) {
      const size_t nt = (size_t)U * max_levels;
      if ((rc = dm_alloc(h, (void **)&d_tc, nt * cap * 4)) != DM_OK) break;
      if ((rc = dm_alloc(h, (void **)&d_ts, nt * cap * 4)) != DM_OK) break;
      if ((rc = dm_alloc(h, (void **)&d_tn, nt * 4)) != DM_OK) break;
      if (hipMemsetAsync(d_tn, 0, nt * 4, h->stream) != hipSuccess || hipMemsetAsync(d_tc, 0, nt * cap * 4, h->stream) != hipSuccess ||
          hipMemsetAsync(d_ts, 0, nt * cap * 4, h->stream) != hipSuccess) { rc = fail(h, DM_ERR_HIP, "memset failed"); break; }
    }
    rc = tdm_search_dev(h, d_seq, U, L, opts, mb, d_coff, d_cids, d_ids, d_scores, d_counts, tn ? max_levels : 0, d_tc, d_ts, d_tn);
    if (rc != DM_OK) break;
    hipError_t e = hipSuccess;
    if (staged) {
      e = hipMemcpyAsync(h->h_stage + b_seq, d_ids, down, hipMemcpyDeviceToHost, h->stream);       // ids | scores | counts are one block of the arena
    } else {
      e = hipMemcpyAsync(out_ids, d_ids, nout * 4, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(out_scores, d_scores, nout * 4, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(out_counts, d_counts, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess && tn) {
      const size_t nt = (size_t)U * max_levels;
      e = hipMemcpyAsync(tc, d_tc, nt * cap * 4, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(ts, d_ts, nt * cap * 4, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(tn, d_tn, nt * 4, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { rc = fail(h, DM_ERR_HIP, std::string("tdm beam search: ") + hipGetErrorString(e)); break; }
    if (staged) {
      memcpy(out_ids, h->h_stage + b_seq, nout * 4);
      memcpy(out_scores, h->h_stage + b_seq + b_out, nout * 4);
      memcpy(out_counts, h->h_stage + b_seq + 2 * b_out, (size_t)U * 4);
    }
  } while (0);
  dm_free_ptr(d_tc); dm_free_ptr(d_tn); dm_free_ptr(d_ts);      // trace buffers (parity instrumentation) are per call
  return rc;
}

int dm_tdm_beam_search(dm_handle_t h, const int32_t *seq_item_ids, int64_t U, int L, const dm_tdm_search_opts *opts,
                       const int64_t *consumed_off, const int32_t *consumed_ids, int32_t *out_item_ids, float *out_scores,
                       int32_t *out_counts) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  return tdm_search_host(h, seq_item_ids, U, L, opts, consumed_off, consumed_ids, out_item_ids, out_scores, out_counts, 0,
                         nullptr, nullptr, nullptr);
}

int dm_tdm_beam_search_trace(dm_handle_t h, const int32_t *seq_item_ids, int64_t U, int L, const dm_tdm_search_opts *opts,
                             int32_t *out_item_ids, float *out_scores, int32_t *out_counts, int max_levels,
                             int32_t *trace_codes, float *trace_scores, int32_t *trace_counts) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  if (max_levels <= 0 || !trace_codes || !trace_scores || !trace_counts) return fail(h, DM_ERR_INVALID, "dm_tdm_beam_search_trace: bad trace arguments");
  return tdm_search_host(h, seq_item_ids, U, L, opts, nullptr, nullptr, out_item_ids, out_scores, out_counts, max_levels,
                         trace_codes, trace_scores, trace_counts);
}

static int otm64_search_dev(dm_ctx *h, const int32_t *d_seq, int64_t U, int L, int beam, int leaf_level, int32_t *d_ids,
                            double *d_sc64, float *d_sc32, int32_t *d_counts, int max_levels, int cap, int32_t *d_tc,
                            double *d_ts64, float *d_ts32, int32_t *d_tn);
static int otm64_search_host(dm_ctx *h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                             int32_t *out_node_ids, double *out_sc64, float *out_sc32, int32_t *out_counts, int max_levels,
                             int32_t *tc, double *ts64, float *ts32, int32_t *tn);

// OTM search on device buffers: d_ids / d_scores [U][2*beam], d_counts [U]; optional level traces (device)
static int otm_search_dev(dm_ctx *h, const int32_t *d_seq, int64_t U, int L, int beam, int leaf_level, int32_t *d_ids,
                          float *d_scores, int32_t *d_counts, int max_levels, int32_t *d_tc, float *d_ts, int32_t *d_tn,
                          const SearchPlan &pl) {
  const int stride = 2 * beam;
  int rc = ensure_ws(h, (size_t)pl.grid * pl.nteams * pl.ws_cap * 16);
  if (rc != DM_OK) return rc;
  HIPCHK(h, hipMemsetAsync(d_ids, 0xFF, (size_t)U * stride * 4, h->stream));
  HIPCHK(h, hipMemsetAsync(d_scores, 0, (size_t)U * stride * 4, h->stream));
  HIPCHK(h, hipMemsetAsync(d_counts, 0, (size_t)U * 4, h->stream));
  if (h->rows_keep) HIPCHK(h, hipMemsetAsync(h->d_rows + 1, 0, 8, h->stream));
  else HIPCHK(h, hipMemsetAsync(h->d_rows, 0, 16, h->stream));
  BeamParams p;
  fill_common(h, p);
  const size_t per = (size_t)pl.grid * pl.nteams * pl.ws_cap;
  p.seq = d_seq; p.U = U; p.L = L; p.use_mask = 1; p.beam = beam; p.topk = stride; p.mode = 1; p.otm_leaf_level = leaf_level;
  p.nteams = pl.nteams; p.cap = pl.cap; p.pcap = pl.pcap; p.out_ids = d_ids; p.out_scores = d_scores; p.out_counts = d_counts; p.out_stride = stride;
  p.ws_code = (int32_t *)h->d_ws; p.ws_score = (float *)h->d_ws + per; p.ws_khi = (uint32_t *)h->d_ws + 2 * per;
  p.ws_klo = (uint32_t *)h->d_ws + 3 * per; p.ws_cap = pl.ws_cap;
  p.trace_codes = d_tc; p.trace_scores = d_ts; p.trace_counts = d_tn; p.trace_levels = d_tn ? max_levels : 0;
  return launch_beam(h, p, pl);
}

static int otm_check(dm_ctx *h, int64_t U, int L, int beam, int leaf_level, const char *who) {
  if (!h->w_loaded) return fail(h, DM_ERR_STATE, std::string(who) + ": weights not loaded");
  if (U < 0 || L <= 0 || L > DM_PIPE_MAXL || beam <= 0 || leaf_level <= 0 || leaf_level > 30)
    return fail(h, DM_ERR_INVALID, std::string(who) + ": bad arguments (L must be 1..32)");
  if ((((int64_t)1) << (leaf_level + 1)) - 1 > h->num_index) return fail(h, DM_ERR_INDEX, std::string(who) + ": leaf level exceeds the embedding table");
  return DM_OK;
}

// device-resident request (serving loops that keep their batches in HBM; the counterpart of dm_tdm_beam_search_dev).
// History codes outside the table are treated as padding by the kernel (the host entry point rejects them).
int dm_otm_beam_search_dev(dm_handle_t h, const int32_t *d_seq_codes, int64_t U, int L, int beam, int leaf_level,
                           int32_t *d_out_node_ids, float *d_out_scores, int32_t *d_out_counts) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  int rc = otm_check(h, U, L, beam, leaf_level, "dm_otm_beam_search_dev");
  if (rc != DM_OK) return rc;
  if (U == 0) return DM_OK;
  if (!d_seq_codes || !d_out_node_ids || !d_out_scores || !d_out_counts) return fail(h, DM_ERR_INVALID, "dm_otm_beam_search_dev: NULL argument");
  HIPCHK(h, hipSetDevice(h->device));
  if (use_f64_beam(h) || long_history_pipeline(h, beam, L)) {      // (long histories beyond the fused kernel's LDS: the per-level pipeline in the model's type)
    return otm64_search_dev(h, d_seq_codes, U, L, beam, leaf_level, d_out_node_ids, nullptr, d_out_scores, d_out_counts, 0, 0, nullptr,
                            nullptr, nullptr, nullptr);
  }
  int start, level;
  level_start_int(beam, &start, &level);
  SearchPlan pl;
  if ((rc = plan_search(h, beam, U, L, leaf_level - level, false, &pl)) != DM_OK) return rc;
  return otm_search_dev(h, d_seq_codes, U, L, beam, leaf_level, d_out_node_ids, d_out_scores, d_out_counts, 0, nullptr, nullptr, nullptr, pl);
}

static int otm_search_host(dm_ctx *h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                           int32_t *out_node_ids, float *out_scores, int32_t *out_counts, int max_levels, int32_t *tc,
                           float *ts, int32_t *tn) {
  int rc = otm_check(h, U, L, beam, leaf_level, "dm_otm_beam_search");
  if (rc != DM_OK) return rc;
  if (U == 0) return DM_OK;          // an empty batch is not an error
  if (!seq_codes || !out_node_ids || !out_scores || !out_counts) return fail(h, DM_ERR_INVALID, "dm_otm_beam_search: bad arguments");
  for (int64_t i = 0; i < U * L; i++)
    if (seq_codes[i] != -1 && (seq_codes[i] < 0 || seq_codes[i] >= h->num_index)) return fail(h, DM_ERR_INDEX, "dm_otm_beam_search: history code outside the embedding table");
  HIPCHK(h, hipSetDevice(h->device));
  int start, level;
  level_start_int(beam, &start, &level);
  SearchPlan pl;
  if ((rc = plan_search(h, beam, U, L, leaf_level - level, false, &pl)) != DM_OK) return rc;
  const size_t stride = (size_t)2 * beam;
  // request arena (grow only): no hipMalloc / hipFree on the request path
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t nt = tn ? (size_t)U * max_levels : 0;
  const size_t o_seq = 0, o_ids = o_seq + up((size_t)U * L * 4), o_sc = o_ids + up(U * stride * 4), o_cnt = o_sc + up(U * stride * 4);
  const size_t o_tc = o_cnt + up((size_t)U * 4), o_ts = o_tc + up(nt * pl.cap * 4), o_tn = o_ts + up(nt * pl.cap * 4);
  const size_t need = o_tn + up(nt * 4);
  if (h->req_bytes < need) {
    dm_free_ptr(h->d_req); h->d_req = nullptr; h->req_bytes = 0;
    if ((rc = dm_alloc(h, &h->d_req, need + need / 2)) != DM_OK) return rc;
    h->req_bytes = need + need / 2;
  }
  char *base = (char *)h->d_req;
  int32_t *d_seq = (int32_t *)(base + o_seq), *d_ids = (int32_t *)(base + o_ids), *d_counts = (int32_t *)(base + o_cnt);
  float *d_scores = (float *)(base + o_sc);
  int32_t *d_tc = tn ? (int32_t *)(base + o_tc) : nullptr, *d_tn = tn ? (int32_t *)(base + o_tn) : nullptr;
  float *d_ts = tn ? (float *)(base + o_ts) : nullptr;
  HIPCHK(h, hipMemcpyAsync(d_seq, seq_codes, (size_t)U * L * 4, hipMemcpyHostToDevice, h->stream));
  int64_t coff_[18];
  const int n_chunks = tn ? 1 : host_pipe_plan(stride * 8, U, coff_);
  if (n_chunks > 1) {      // chunks of users, downloads under the kernels behind them (host_pipe_plan)
    if ((rc = host_pipe_ensure(h, n_chunks)) != DM_OK) return rc;
    int launched = 0;
    for (int k = 0; k < n_chunks && rc == DM_OK; k++) {
      const int64_t u0 = coff_[k], uk = coff_[k + 1] - u0;
      if (uk <= 0) break;
      SearchPlan plk;
      if ((rc = plan_search(h, beam, uk, L, leaf_level - level, false, &plk)) != DM_OK) break;
      h->rows_keep = k > 0;
      rc = otm_search_dev(h, d_seq + u0 * L, uk, L, beam, leaf_level, d_ids + u0 * stride, d_scores + u0 * stride, d_counts + u0, 0, nullptr, nullptr, nullptr, plk);
      h->rows_keep = false;
      if (rc == DM_OK && hipEventRecord(h->chunk_ev[(size_t)k], h->stream) != hipSuccess) rc = fail(h, DM_ERR_HIP, "dm_otm_beam_search: event record failed");
      if (rc == DM_OK) launched++;
    }
    hipError_t e = hipSuccess;
    for (int k = 0; k < launched && e == hipSuccess; k++) {
      const int64_t u0 = coff_[k], uk = coff_[k + 1] - u0;
      e = hipStreamWaitEvent(h->copy_stream, h->chunk_ev[(size_t)k], 0);
      if (e == hipSuccess) e = hipMemcpyAsync(out_node_ids + u0 * stride, d_ids + u0 * stride, (size_t)uk * stride * 4, hipMemcpyDeviceToHost, h->copy_stream);
      if (e == hipSuccess) e = hipMemcpyAsync(out_scores + u0 * stride, d_scores + u0 * stride, (size_t)uk * stride * 4, hipMemcpyDeviceToHost, h->copy_stream);
      if (e == hipSuccess) e = hipMemcpyAsync(out_counts + u0, d_counts + u0, (size_t)uk * 4, hipMemcpyDeviceToHost, h->copy_stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->copy_stream);
    if (rc != DM_OK) { (void)hipStreamSynchronize(h->stream); return rc; }
    if (e != hipSuccess) { (void)hipStreamSynchronize(h->stream); return fail(h, DM_ERR_HIP, std::string("dm_otm_beam_search (pipelined download): ") + hipGetErrorString(e)); }
    HIPCHK(h, hipMemcpyAsync(&h->h_rows, h->d_rows, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->last_rows = (int64_t)h->h_rows;
    return DM_OK;
  }
  if (tn) HIPCHK(h, hipMemsetAsync(d_tn, 0, nt * 4, h->stream));
  if ((rc = otm_search_dev(h, d_seq, U, L, beam, leaf_level, d_ids, d_scores, d_counts, max_levels, d_tc, d_ts, d_tn, pl)) != DM_OK) return rc;
  if (tn) {
    HIPCHK(h, hipMemcpyAsync(tc, d_tc, nt * pl.cap * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ts, d_ts, nt * pl.cap * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(tn, d_tn, nt * 4, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipMemcpyAsync(out_node_ids, d_ids, U * stride * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(out_scores, d_scores, U * stride * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(out_counts, d_counts, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&h->h_rows, h->d_rows, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->last_rows = (int64_t)h->h_rows;
  return DM_OK;
}

int dm_otm_beam_search(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                       int32_t *out_node_ids, float *out_scores, int32_t *out_counts) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  if (use_f64_beam(h) || long_history_pipeline(h, beam, L))      // f64 weights: the reference's arithmetic (otm64.hip.inc), scores rounded to float on the way out
    return otm64_search_host(h, seq_codes, U, L, beam, leaf_level, out_node_ids, nullptr, out_scores, out_counts, 0, nullptr, nullptr, nullptr, nullptr);
  return otm_search_host(h, seq_codes, U, L, beam, leaf_level, out_node_ids, out_scores, out_counts, 0, nullptr, nullptr, nullptr);
}

int dm_otm_beam_search_trace(dm_handle_t h, const int32_t *seq_codes, int64_t U, int L, int beam, int leaf_level,
                             int32_t *out_node_ids, float *out_scores, int32_t *out_counts, int max_levels,
                             int32_t *trace_codes, float *trace_scores, int32_t *trace_counts) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  if (max_levels <= 0 || !trace_codes || !trace_scores || !trace_counts) return fail(h, DM_ERR_INVALID, "dm_otm_beam_search_trace: bad trace arguments");
  if (use_f64_beam(h) || long_history_pipeline(h, beam, L))
    return otm64_search_host(h, seq_codes, U, L, beam, leaf_level, out_node_ids, nullptr, out_scores, out_counts, max_levels, trace_codes,
                             nullptr, trace_scores, trace_counts);
  return otm_search_host(h, seq_codes, U, L, beam, leaf_level, out_node_ids, out_scores, out_counts, max_levels, trace_codes,
                         trace_scores, trace_counts);
}


// Brute force over every leaf with the same fused scorer (mode 2 of the beam kernel): the
// build-defined oracle for recall@k.  Order: score descending, then leaf code ascending.
int dm_tdm_bruteforce_topk(dm_handle_t h, const int32_t *seq_item_ids, int64_t U, int L, int topk, int use_mask,
                           int32_t *out_item_ids, float *out_scores, int32_t *out_counts) {
  if (!h) return DM_ERR_INVALID;
  DM_CLONE_ENTER(h);
  if (!h->tree_loaded || !h->ids_loaded || !h->w_loaded) return fail(h, DM_ERR_STATE, "dm_tdm_bruteforce_topk: tree, id maps and weights must be loaded first");
  if (U == 0 && L > 0 && L <= DM_PIPE_MAXL && topk > 0 && topk <= 256) return DM_OK;          // an empty batch is not an error
  if (!seq_item_ids || !out_item_ids || !out_scores || !out_counts || U <= 0 || L <= 0 || L > DM_PIPE_MAXL || topk <= 0 || topk > 256)
    return fail(h, DM_ERR_INVALID, "dm_tdm_bruteforce_topk: bad arguments (L must be 1..32, topk 1..256)");
  if (h->n_slots > h->num_index) return fail(h, DM_ERR_INDEX, "dm_tdm_bruteforce_topk: tree codes exceed the embedding table");
  HIPCHK(h, hipSetDevice(h->device));
  const int pcap = 512;
  const int chunk = ((pcap - topk) / 16) * 16;
  const int cap = chunk < 32 ? 32 : chunk;
  const int kt = L > DM_MAXL ? 2 : 1;
  const int kq = kt > 1 ? 4 : (L + 3) / 4;
  int nteams = 0, lds = 0;
  for (int cand = 4; cand >= 1; cand >>= 1) {
    BeamLds l = dm_beam_lds(h->embed, cand, cap, pcap, kq, false, kt);
    if (l.total <= 160 * 1024) { nteams = cand; lds = l.total; break; }
  }
  if (!nteams) return fail(h, DM_ERR_UNSUPPORTED, "dm_tdm_bruteforce_topk: LDS budget exceeded");
  // enough (user, slice) work items to fill every team a few times over
  int64_t slices = (4 * (int64_t)h->n_cu * nteams + U - 1) / U;
  const int64_t max_slices = (h->n_leaf_nodes + chunk - 1) / chunk;
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  int64_t per = (h->n_leaf_nodes + slices - 1) / slices;
  per = ((per + 15) / 16) * 16;
  slices = (h->n_leaf_nodes + per - 1) / per;
  const int64_t n_work = U * slices;
  if (n_work >= (1ll << 31)) return fail(h, DM_ERR_UNSUPPORTED, "dm_tdm_bruteforce_topk: too many work items");
  int32_t *d_seq = nullptr;
  unsigned long long *d_keys = nullptr;
  int rc = DM_OK;
  std::vector<unsigned long long> keys((size_t)n_work * topk);
  do {
    if ((rc = ensure_ws(h, 1024)) != DM_OK) break;
    if ((rc = dm_alloc(h, (void **)&d_seq, (size_t)U * L * 4)) != DM_OK) break;
    if ((rc = dm_alloc(h, (void **)&d_keys, keys.size() * 8)) != DM_OK) break;
    hipError_t e = hipMemcpyAsync(d_seq, seq_item_ids, (size_t)U * L * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_rows, 0, 16, h->stream);
    if (e != hipSuccess) { rc = fail(h, DM_ERR_HIP, "dm_tdm_bruteforce_topk: upload failed"); break; }
    BeamParams p;
    fill_common(h, p);
    p.seq = d_seq; p.U = U; p.L = L; p.use_mask = use_mask; p.beam = 2; p.topk = topk; p.mode = 2;
    p.nteams = nteams; p.cap = cap; p.pcap = pcap; p.out_stride = topk;
    p.bf_leaf_codes = h->d_leaf_codes; p.bf_n_leaf = h->n_leaf_nodes; p.bf_slices = (int)slices; p.bf_chunk = chunk;
    p.bf_per_slice = per; p.bf_out_keys = d_keys;
    p.ws_code = (int32_t *)h->d_ws; p.ws_score = (float *)h->d_ws; p.ws_khi = (uint32_t *)h->d_ws; p.ws_klo = (uint32_t *)h->d_ws;
    p.ws_cap = 0;
    SearchPlan pl;
    pl.nteams = nteams; pl.cap = cap; pl.pcap = pcap; pl.lds = lds; pl.ws_cap = 0; pl.wkernel = false;
    int64_t groups = (n_work + nteams - 1) / nteams;
    pl.grid = (int)(groups < h->n_cu ? groups : h->n_cu);
    if ((rc = launch_beam(h, p, pl)) != DM_OK) break;
    e = hipMemcpyAsync(keys.data(), d_keys, keys.size() * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { rc = fail(h, DM_ERR_HIP, std::string("dm_tdm_bruteforce_topk: ") + hipGetErrorString(e)); break; }
    std::vector<int32_t> nid((size_t)h->n_slots);
    if (hipMemcpy(nid.data(), h->d_node_id, nid.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(h, DM_ERR_HIP, "node id download failed"); break; }
    // merge the per-slice winners (keys are unique: descending-score key << 32 | leaf code)
    std::vector<unsigned long long> row((size_t)slices * topk);
    for (int64_t u = 0; u < U; u++) {
      std::copy(keys.begin() + u * slices * topk, keys.begin() + (u + 1) * slices * topk, row.begin());
      const size_t kk = std::min<size_t>(topk, row.size());
      std::partial_sort(row.begin(), row.begin() + kk, row.end());
      int n = 0;
      for (size_t i = 0; i < kk; i++) {
        if (row[i] == ~0ull) break;
        const uint32_t code = (uint32_t)row[i], dk = (uint32_t)(row[i] >> 32);
        const uint32_t asc = ~dk;
        const uint32_t bits = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
        float sc; memcpy(&sc, &bits, 4);
        out_item_ids[u * topk + n] = nid[code];
        out_scores[u * topk + n] = sc;
        n++;
      }
      for (int i = n; i < topk; i++) { out_item_ids[u * topk + i] = -1; out_scores[u * topk + i] = 0.f; }
      out_counts[u] = n;
    }
  } while (0);
  dm_free_ptr(d_seq); dm_free_ptr(d_keys);
  return rc;
}

#include "jtm_rebalance_dev.hip.inc"
#include "jtm_host.hip.inc"
#include "train_host.hip.inc"
#include "sampler.hip.inc"
#include "dr_host.hip.inc"
#include "otm64.hip.inc"
#include "tdm_pipeline.hip.inc"
#include "comm.hip.inc"
#include "jtm_sharded.hip.inc"
#include "train_grouped_host.hip.inc"
#include "otm_train.hip.inc"
#include "checkpoint.hip.inc"
#include "tree_file.hip.inc"

// ---- device memory helpers
int dm_dev_alloc(dm_handle_t h, size_t bytes, void **dptr) {
  if (!h || !dptr) return DM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  return dm_alloc(h, dptr, bytes);
}
int dm_dev_free(dm_handle_t h, void *dptr) {
  if (!h) return DM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (dptr) HIPCHK(h, hipFree(dptr));
  return DM_OK;
}
int dm_memcpy_h2d(dm_handle_t h, void *dst, const void *src, size_t bytes) {
  if (!h) return DM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return DM_OK;
}
int dm_memcpy_d2h(dm_handle_t h, void *dst, const void *src, size_t bytes) {
  if (!h) return DM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return DM_OK;
}

// ---- measurement
int dm_kernel_timing_reset(dm_handle_t h) {
  if (!h) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->ev_used = 0;
  return DM_OK;
}
int dm_kernel_timing_get(dm_handle_t h, int *launches, double *total_ms) {
  if (!h || !launches || !total_ms) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double tot = 0;
  int n = 0;
  for (size_t i = 0; i < h->ev_used; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second) != hipSuccess) { (void)hipGetLastError(); continue; }     // (see dm_kernel_timing_get_kind)
    tot += ms; n++;
  }
  *launches = n; *total_ms = tot;
  return DM_OK;
}
const char *dm_last_beam_kernel(dm_handle_t h) { return h ? h->last_kernel : ""; }
int dm_kernel_timing_get_kind(dm_handle_t h, int kind, int *launches, double *total_ms) {
  if (!h || !launches || !total_ms) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double tot = 0;
  int n = 0;
  for (size_t i = 0; i < h->ev_used; i++) {
    if (h->ev_kind[i] != kind) continue;
    float ms = 0;
    // (a pair whose stop event was never recorded — a call that failed between its two records — is skipped, not an error of this query)
    if (hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second) != hipSuccess) { (void)hipGetLastError(); continue; }
    tot += ms; n++;
  }
  *launches = n; *total_ms = tot;
  return DM_OK;
}
// debug (not part of the public header): cumulative per-phase wave cycles, DM_PHASE_TIMERS builds
extern "C" int dm_debug_phase_cycles(dm_handle_t h, unsigned long long *out8) {
  if (!h || !out8 || !h->d_phase) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out8, h->d_phase, 128, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemset(h->d_phase, 0, 128));
  return DM_OK;
}
// debug (not part of the public header): Deep-Retrieval layers that fell back to the exact radix-select path
extern "C" int dm_debug_dr_slow_layers(dm_handle_t h, unsigned long long *out, int reset) {
  if (!h || !out) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_rows + 2, 8, hipMemcpyDeviceToHost));
  if (reset) HIPCHK(h, hipMemset(h->d_rows + 2, 0, 8));
  return DM_OK;
}
// debug: user-layers the one-wave cut of the sliced Deep-Retrieval search handed to the block version
extern "C" int dm_debug_dr_wave_fallbacks(dm_handle_t h, unsigned long long *out, int reset) {
  if (!h || !out) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_rows + 5, 16, hipMemcpyDeviceToHost));      // out[0] = count, out[1] = per-reason byte counters
  if (reset) HIPCHK(h, hipMemset(h->d_rows + 5, 0, 16));
  return DM_OK;
}
int dm_last_scored_rows(dm_handle_t h, int64_t *rows) {
  if (!h || !rows) return DM_ERR_INVALID;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unsigned long long r = 0;
  HIPCHK(h, hipMemcpy(&r, h->d_rows, 8, hipMemcpyDeviceToHost));
  h->last_rows = (int64_t)r;
  *rows = h->last_rows;
  return DM_OK;
}

