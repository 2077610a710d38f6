// dismember.hpp — C++17 host side above the C ABI (include/dismember_hip.h), header-only.
//
// The reference's host code is compiled JVM code (Scala); this mirrors its public interface for the retrieval path
// in C++ — same class / method names and argument meaning — so that a C++ service (or the JNI shim of
// INTEGRATION.md) gets the reference's API without Python:
//
//   dm::TDM::recommend / recommendItems     tdm/src/main/scala/com/mass/tdm/model/TDM.scala:17-22,
//                                           tdm/.../model/Recommender.scala:18-37
//   dm::OTM::recommend                      otm/src/main/scala/com/mass/otm/model/OTM.scala:14-22
//   dm::DeepRetrieval::recommend            deep-retrieval/.../model/DeepRetrieval.scala:26-46
//   dm::JTM::optimize                       jtm/src/main/scala/com/mass/jtm/optim/JTM.scala:22-73
//   dm::Metrics::computeMetrics             tdm/.../evaluation/Metrics.scala:5-25
//
// Errors of the C ABI become dm::Error (code = dm_status, what() = dm_last_error), the counterpart of the exceptions
// the Scala code throws (`require`, ArrayIndexOutOfBoundsException of LookupTable.scala:47-53).
// Every method also has a batch form ([U x L] row-major sequences): one device call per batch is what the GPU is for.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <map>
#include <thread>
#include <stdexcept>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "dismember_hip.h"

namespace dm {

// Property.readConf / getOrStop / getCoreNumber (scalann/src/main/scala/com/mass/scalann/utils/Property.scala:12-71): the
// reference's `.conf` format, verbatim — lines `prefix.key<whitespace>value`, kept when they START with the prefix and split
// into exactly two tokens; later duplicates win — so configs/*.conf of the reference load unchanged.
struct Property {
  static std::map<std::string, std::string> readConf(const std::string &path, const std::string &prefix, bool truncate = true) {
    std::ifstream f(path);
    if (!f) throw std::invalid_argument("requirement failed: Config file " + path + " doesn't exist");
    std::map<std::string, std::string> out;
    std::string line;
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\x0b' || c == '\f' || c == '\r'; };
    while (std::getline(f, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.compare(0, prefix.size(), prefix) != 0) continue;
      size_t a = 0, b = line.size();
      while (a < b && (unsigned char)line[a] <= ' ') a++;            // String.trim
      while (b > a && (unsigned char)line[b - 1] <= ' ') b--;
      std::vector<std::string> tok;
      for (size_t i = a; i < b;) {
        size_t j = i;
        while (j < b && !ws(line[j])) j++;
        tok.push_back(line.substr(i, j - i));
        while (j < b && ws(line[j])) j++;
        i = j;
      }
      if (tok.size() != 2) continue;
      out[truncate ? (tok[0].size() > prefix.size() ? tok[0].substr(prefix.size() + 1) : std::string()) : tok[0]] = tok[1];
    }
    return out;
  }
  static const std::string &getOrStop(const std::map<std::string, std::string> &conf, const std::string &key) {
    auto it = conf.find(key);
    if (it == conf.end()) throw std::invalid_argument("failed to read parameter: " + key + " in conf file");
    return it->second;
  }
  static int getCoreNumber(int confNum) {           // thread_number <= 0: every available processor
    if (confNum > 0) return confNum;
    const unsigned n = std::thread::hardware_concurrency();
    return n ? (int)n : 1;
  }
};

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &msg) : std::runtime_error("dismember_hip error " + std::to_string(c) + ": " + msg), code(c) {}
};

// one dm_handle_t = one device + stream; not copyable
class Engine {
  explicit Engine(dm_handle_t adopted) : h_(adopted) {}      // cloneEngine()
 public:
  explicit Engine(int device_id = 0) {
    const int rc = dm_create(device_id, &h_);
    if (rc != DM_OK) throw Error(rc, dm_last_error(nullptr) ? dm_last_error(nullptr) : "");
  }
  ~Engine() { if (h_) dm_destroy(h_); }
  // Module.cloneModule() as the reference's workers use it (tdm/.../optim/LocalOptimizer.scala:28-44; otm/src/test/scala/
  // CloneModelSpec.scala:20-36): a second engine that READS this engine's tree and weights (dm_clone: nothing is copied) through its
  // own stream and buffers — one per serving thread.  Loading and training stay with the owner; destroy the clones first.
  std::unique_ptr<Engine> cloneEngine() const {
    dm_handle_t c = nullptr;
    check(dm_clone(h_, &c));
    std::unique_ptr<Engine> e(new Engine(c));
    e->maxLevel_ = maxLevel_; e->embed_ = embed_;
    return e;
  }
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;
  dm_handle_t handle() const { return h_; }
  void check(int rc) const { if (rc != DM_OK) throw Error(rc, dm_last_error(h_) ? dm_last_error(h_) : ""); }

  // DistTree.loadData result (tdm/.../tree/DistTree.scala:40-87) -> device index
  void loadTree(const std::vector<int32_t> &codes, const std::vector<int32_t> &nodeIds, const std::vector<uint8_t> &isLeaf,
                int maxLevel, const std::vector<int32_t> &leafItemIds, const std::vector<int32_t> &leafCodes) {
    check(dm_load_tree_tdm(h_, codes.data(), nodeIds.data(), isLeaf.data(), (int64_t)codes.size(), maxLevel));
    check(dm_load_id_maps(h_, leafItemIds.data(), leafCodes.data(), (int64_t)leafItemIds.size()));
    maxLevel_ = maxLevel;
  }
  // the compact parameter vector of Module.parameters() (scalann/.../nn/graphnn/Graph.scala:37-48)
  void loadWeightsDin(const std::vector<float> &compact, int embedSize, int64_t numIndex) {
    check(dm_load_weights_din(h_, DM_F32, embedSize, numIndex, compact.data(), (int64_t)compact.size()));
    embed_ = embedSize;
  }
  void loadWeightsDin(const std::vector<double> &compact, int embedSize, int64_t numIndex) {
    check(dm_load_weights_din(h_, DM_F64, embedSize, numIndex, compact.data(), (int64_t)compact.size()));
    embed_ = embedSize;
  }
  // TDM.saveModel / TDM.loadModel (tdm/.../model/TDM.scala:32-54): weights + index in one flat file (dm_save_model)
  void saveModel(const std::string &path) const { check(dm_save_model(h_, path.c_str())); }
  void loadModel(const std::string &path) { check(dm_load_model(h_, path.c_str())); }
  // arithmetic of the beam-search scorer: DM_SCORER_AUTO (default), DM_SCORER_F32, DM_SCORER_SPLIT_F16 (dismember_hip.h)
  void setScorerMode(int mode) { check(dm_set_scorer_mode(h_, mode)); }
  int scorerModeInEffect() const {
    int m = 0, eff = 0;
    dm_get_scorer_mode(h_, &m, &eff, nullptr, nullptr);
    return eff;
  }
  int maxLevel() const { return maxLevel_; }
  int embedSize() const { return embed_; }
  // multi-GPU: the communicator this engine's collectives use (dm_comm_create_tcp / _rccl / _all); not owned, nullptr detaches
  void attachComm(dm_comm_t c) { check(dm_comm_attach(h_, c)); }

 private:
  dm_handle_t h_ = nullptr;
  int maxLevel_ = 0, embed_ = 0;
};

inline double sigmoid(double logit) { return 1.0 / (1.0 + std::exp(-logit)); }   // TDM.sigmoid, in double (TDM.scala:56-58)

using Recs = std::vector<std::pair<int, double>>;   // Array[(Int, Double)]

class TDM {
 public:
  TDM(Engine &engine, const std::string &modelName = "din") : e_(engine) {
    std::string m = modelName;
    std::transform(m.begin(), m.end(), m.begin(), ::tolower);
    useMask_ = (m == "din");                         // TDM.apply, TDM.scala:26-29
  }
  // TDM.predict(sequence, target): Double (TDM.scala:10-15): idToCode over sequence ++ target, one forward, sigmoid in double
  double predict(const std::vector<int32_t> &sequence, int32_t target) const {
    std::vector<int32_t> all(sequence);
    all.push_back(target);
    std::vector<int32_t> codes(all.size()), maskPos(all.size());
    int nMask = 0;
    e_.check(dm_tdm_id_to_code(e_.handle(), all.data(), (int)all.size(), codes.data(), maskPos.data(), &nMask));
    std::vector<int32_t> pad;
    if (useMask_) for (int i = 0; i < nMask; i++) if (maskPos[(size_t)i] < (int32_t)sequence.size()) pad.push_back(maskPos[(size_t)i]);
    float logit = 0.f;
    const int32_t dummy = 0;
    e_.check(dm_din_forward(e_.handle(), &codes.back(), codes.data(), pad.empty() ? &dummy : pad.data(), (int64_t)pad.size(), 1, (int)sequence.size(), &logit));
    return sigmoid((double)logit);
  }
  // TDM.saveModel / loadModel / loadTree companions (TDM.scala:32-54): one flat checkpoint; the reference's own tree file
  static void saveModel(const std::string &modelPath, const Engine &engine) { engine.saveModel(modelPath); }
  static TDM loadModel(Engine &engine, const std::string &modelPath, const std::string &modelName = "din") {
    engine.loadModel(modelPath);
    return TDM(engine, modelName);
  }
  static void loadTree(Engine &engine, const std::string &treePbPath) { engine.check(dm_load_tree_file(engine.handle(), treePbPath.c_str())); }
  // TDM.recommend(sequence, topk, candidateNum): Array[(Int, Double)]
  Recs recommend(const std::vector<int32_t> &sequence, int topk, int candidateNum) const {
    return recommendBatch(sequence, 1, (int)sequence.size(), topk, candidateNum)[0];
  }
  std::vector<Recs> recommendBatch(const std::vector<int32_t> &sequences, int64_t U, int L, int topk, int candidateNum) const {
    std::vector<int32_t> ids((size_t)U * topk), cnt((size_t)U);
    std::vector<float> sc((size_t)U * topk);
    dm_tdm_search_opts o{candidateNum, topk, useMask_ ? 1 : 0, 0};
    e_.check(dm_tdm_beam_search(e_.handle(), sequences.data(), U, L, &o, nullptr, nullptr, ids.data(), sc.data(), cnt.data()));
    std::vector<Recs> out((size_t)U);
    for (int64_t u = 0; u < U; u++)
      for (int i = 0; i < cnt[(size_t)u]; i++)
        out[(size_t)u].emplace_back(ids[(size_t)(u * topk + i)], sigmoid((double)sc[(size_t)(u * topk + i)]));
    return out;
  }
  // Recommender.recommendItems(sequence, ..., consumedItems: Option[Seq[Int]]): ids only; with consumed items the beam
  // is widened to max(candidateNum, (consumed + topk) / 2) and the consumed ids are dropped (Recommender.scala:28-36)
  std::vector<int32_t> recommendItems(const std::vector<int32_t> &sequence, int topk, int candidateNum,
                                      const std::vector<int32_t> *consumedItems = nullptr) const {
    std::vector<int32_t> ids((size_t)topk), cnt(1);
    std::vector<float> sc((size_t)topk);
    dm_tdm_search_opts o{candidateNum, topk, useMask_ ? 1 : 0, consumedItems ? 1 : 0};
    const int64_t off[2] = {0, consumedItems ? (int64_t)consumedItems->size() : 0};
    const int32_t dummy = 0;
    e_.check(dm_tdm_beam_search(e_.handle(), sequence.data(), 1, (int)sequence.size(), &o, consumedItems ? off : nullptr,
                                consumedItems ? (consumedItems->empty() ? &dummy : consumedItems->data()) : nullptr,
                                ids.data(), sc.data(), cnt.data()));
    ids.resize((size_t)cnt[0]);
    return ids;
  }

 private:
  Engine &e_;
  bool useMask_;
};

class OTM {
 public:
  // itemIdMapping: item -> leaf node id (OTM.scala:8-12)
  OTM(Engine &engine, const std::map<int32_t, int32_t> &itemIdMapping) : e_(engine), itemId_(itemIdMapping) {
    for (auto &kv : itemId_) idItem_[kv.second] = kv.first;
    leafLevel_ = (int)std::ceil(std::log((double)itemId_.size()) / std::log(2.0));   // upperLog2, otm/package.scala:16
  }
  // OTM.recommend(sequence, topk, beamSize): Seq[(Int, Double)] (OTM.scala:14-22)
  Recs recommend(const std::vector<int32_t> &sequence, int topk, int beamSize) const {
    const int L = (int)sequence.size();
    std::vector<int32_t> codes((size_t)L);
    for (int j = 0; j < L; j++) {
      auto it = itemId_.find(sequence[(size_t)j]);
      codes[(size_t)j] = it == itemId_.end() ? -1 : it->second;
    }
    std::vector<int32_t> nodes((size_t)2 * beamSize), cnt(1);
    std::vector<float> sc((size_t)2 * beamSize);
    e_.check(dm_otm_beam_search(e_.handle(), codes.data(), 1, L, beamSize, leafLevel_, nodes.data(), sc.data(), cnt.data()));
    std::vector<std::pair<int32_t, double>> keep;
    for (int i = 0; i < cnt[0]; i++) {
      auto it = idItem_.find(nodes[(size_t)i]);            // filter(idItemMapping.contains)
      if (it != idItem_.end()) keep.emplace_back(it->second, (double)sc[(size_t)i]);
    }
    std::stable_sort(keep.begin(), keep.end(), [](const auto &a, const auto &b) { return a.second > b.second; });   // sortBy(_.score)(reverse)
    Recs out;
    for (size_t i = 0; i < keep.size() && (int)i < topk; i++) out.emplace_back(keep[i].first, sigmoid(keep[i].second));
    return out;
  }
  int leafLevel() const { return leafLevel_; }

 private:
  Engine &e_;
  std::map<int32_t, int32_t> itemId_;
  std::unordered_map<int32_t, int32_t> idItem_;
  int leafLevel_ = 0;
};

// OTM training: LocalOptimizer's iteration (otm/src/main/scala/com/mass/otm/optim/LocalOptimizer.scala:55-109) for one worker's batch
// as one library call (dm_otm_train_batch): pseudo targets, beam nodes and the per-level label join on the device, one forward /
// backward + gradient exchange (when the engine carries a communicator with > 1 rank: every rank calls trainBatch) + Adam per level.
class OTMLocalOptimizer {
 public:
  OTMLocalOptimizer(Engine &engine, int leafLevel, int beamSize, int seqLen, double learningRate, bool useMask = true,
                    const std::string &targetMode = "pseudo")
      : e_(engine), L_(seqLen) {
    opts_.beam = beamSize; opts_.leaf_level = leafLevel; opts_.use_mask = useMask ? 1 : 0; opts_.target_mode = targetMode == "normal" ? 1 : 0;
    dm_adam_opts a{learningRate, 0.0, 0.9, 0.999, 1e-8};                 // Adam defaults, scalann/.../optim/Adam.scala:10-16
    e_.check(dm_train_init(e_.handle(), &a));
  }
  // sequences [U x seqLen] node ids (-1 = padding), targetNodes[u] = the user's target leaf nodes; -> the per-level losses
  std::vector<double> trainBatch(const std::vector<int32_t> &sequences, const std::vector<std::vector<int32_t>> &targetNodes) {
    const int64_t U = (int64_t)targetNodes.size();
    if ((int64_t)sequences.size() != U * L_) throw Error(DM_ERR_INVALID, "OTMLocalOptimizer: sequences must hold U x seqLen ids");
    std::vector<int64_t> off((size_t)U + 1, 0);
    std::vector<int32_t> flat;
    for (int64_t u = 0; u < U; u++) { flat.insert(flat.end(), targetNodes[(size_t)u].begin(), targetNodes[(size_t)u].end()); off[(size_t)u + 1] = (int64_t)flat.size(); }
    if (flat.empty()) flat.push_back(0);
    std::vector<double> losses(64, 0.0);
    int n = 0;
    e_.check(dm_otm_train_batch(e_.handle(), sequences.data(), U, L_, off.data(), flat.data(), &opts_, losses.data(), &n));
    losses.resize((size_t)n);
    return losses;
  }

 private:
  Engine &e_;
  int L_;
  dm_otm_train_opts opts_{};
};

class DeepRetrieval {
 public:
  // itemIdMapping: item -> internal id (MappingOp.itemIdMapping); model and path table are loaded into the engine with
  // dm_dr_load_model / dm_dr_load_path_items
  DeepRetrieval(Engine &engine, const std::map<int32_t, int32_t> &itemIdMapping) : e_(engine), itemId_(itemIdMapping) {
    for (auto &kv : itemId_) idItem_[kv.second] = kv.first;
  }
  // DeepRetrieval.recommend(sequence, topk, beamSize, mappings): Seq[(Int, Double)]
  Recs recommend(const std::vector<int32_t> &sequence, int topk, int beamSize) const {
    std::vector<int32_t> ids(sequence.size());
    for (size_t j = 0; j < sequence.size(); j++) {
      auto it = itemId_.find(sequence[j]);
      ids[j] = it == itemId_.end() ? -1 : it->second;        // getOrElse(_, paddingIdx)
    }
    std::vector<int32_t> out((size_t)topk), cnt(1);
    std::vector<double> sc((size_t)topk);
    e_.check(dm_dr_recommend(e_.handle(), ids.data(), 1, beamSize, topk, out.data(), sc.data(), cnt.data()));
    Recs r;
    for (int i = 0; i < cnt[0]; i++) r.emplace_back(idItem_.at(out[(size_t)i]), sigmoid(sc[(size_t)i]));
    return r;
  }

 private:
  Engine &e_;
  std::map<int32_t, int32_t> itemId_;
  std::unordered_map<int32_t, int32_t> idItem_;
};

struct Metrics {
  // Metrics.computeMetrics(recItems, labels): (precision, recall, ndcg)
  static void computeMetrics(const std::vector<int32_t> &recItems, const std::vector<int32_t> &labels, double &precision,
                             double &recall, double &ndcg) {
    const std::unordered_set<int32_t> labelSet(labels.begin(), labels.end());
    int common = 0, j = 0;
    double dcg = 0.0, idcg = 0.0;
    for (size_t i = 0; i < recItems.size(); i++)
      if (labelSet.count(recItems[i])) {
        common++;
        dcg += std::log(2.0) / std::log((double)i + 2.0);
        idcg += std::log(2.0) / std::log((double)j + 2.0);
        j++;
      }
    if (common) { precision = (double)common / (double)recItems.size(); recall = (double)common / (double)labels.size(); ndcg = dcg / idcg; }
    else precision = recall = ndcg = 0.0;
  }
};

// JTM tree learning: JTM.optimize (JTM.scala:22-73) as one library call (dm_jtm_optimize_cached).  The reference's `numThreads`
// workers (JTM.scala:33-68) are GPUs here: attach a communicator to the engine (Engine::attachComm: one process per GPU, every process
// constructs the same JTM and calls optimize() — collective), or hand optimizeAll() the engines of one process driving several GPUs
// (dm_comm_create_all clique); both give the single-GPU projection bit for bit.
class JTM {
 public:
  // leafItemIds / leafCodes: the CURRENT tree's item -> leaf code map; itemRows[item] = flattened [rows x seqLen] histories
  // (itemSequenceMap, TreeLearning.scala:34-46).  Items are visited in ascending id (the reference iterates a HashMap).
  JTM(Engine &engine, const std::vector<int32_t> &leafItemIds, const std::vector<int32_t> &leafCodes, int maxLevel,
      const std::map<int32_t, std::vector<int32_t>> &itemRows, int gap = 2, int seqLen = 10, bool hierarchical = false,
      int minLevel = 0, bool useMask = true)
      : e_(engine), maxLevel_(maxLevel), gap_(gap), L_(seqLen), hier_(hierarchical), minLevel_(minLevel), useMask_(useMask) {
    std::vector<std::pair<int32_t, int32_t>> ic(leafItemIds.size());
    for (size_t i = 0; i < ic.size(); i++) ic[i] = {leafItemIds[i], leafCodes[i]};
    std::sort(ic.begin(), ic.end());
    rowOff_.assign(ic.size() + 1, 0);
    for (size_t k = 0; k < ic.size(); k++) {
      items_.push_back(ic[k].first);
      itemCode_.push_back(ic[k].second);
      auto it = itemRows.find(ic[k].first);
      const size_t n = it == itemRows.end() ? 0 : it->second.size();
      if (n % (size_t)L_) throw Error(DM_ERR_INVALID, "JTM: item rows must be multiples of seqLen");
      if (n) rowIds_.insert(rowIds_.end(), it->second.begin(), it->second.end());
      rowOff_[k + 1] = rowOff_[k] + (int64_t)(n / (size_t)L_);
    }
    if (rowIds_.empty()) rowIds_.assign((size_t)L_, 0);
  }
  static int32_t ancestorAtLevel(int32_t code, int level) {      // JTMTree.getAncestorAtLevel (JTMTree.scala:36-43)
    const int64_t lim = ((int64_t)1 << (level + 1)) - 1;
    int64_t c = code;
    while (c >= lim) c = (c - 1) >> 1;
    return (int32_t)c;
  }
  // -> item id -> new leaf code
  std::map<int32_t, int32_t> optimize() {
    const size_t n = items_.size();
    std::vector<int32_t> proj(n, 0);                               // first all assigned to the root (:23-26)
    // itemSequenceMap goes to the device once; every gap step is one call (scoring + greedy re-balance, weights stay in HBM)
    e_.check(dm_jtm_cache_rows(e_.handle(), rowOff_.data(), rowIds_.data(), (int64_t)n, L_));
    struct Drop { Engine &e; int L; ~Drop() { dm_jtm_cache_rows(e.handle(), nullptr, nullptr, 0, L); } } drop{e_, L_};
    // the loop over the gap steps is one call: projection and weights stay in HBM between the steps; dropped items keep their old node (:72)
    e_.check(dm_jtm_optimize_cached(e_.handle(), itemCode_.data(), (int64_t)n, maxLevel_, gap_, hier_ ? 1 : 0, minLevel_, useMask_ ? 1 : 0,
                                    proj.data(), nullptr));
    std::map<int32_t, int32_t> res;
    for (size_t i = 0; i < n; i++) res[items_[i]] = proj[i];
    return res;
  }
  // one process, several GPUs: engines[i] carries rank i of a dm_comm_create_all clique (engines[0] may be this object's engine);
  // every engine gets the rows of ITS item range, the ranks run on host threads inside the call and must end with the same projection
  std::map<int32_t, int32_t> optimizeAll(const std::vector<Engine *> &engines) {
    const size_t n = items_.size();
    std::vector<int32_t> proj(n, 0);
    std::vector<dm_handle_t> hs;
    for (size_t r = 0; r < engines.size(); r++) {             // every rank: the bookkeeping of all items, the ROWS of the range it scores
      int64_t lo = 0, hi = 0;
      engines[r]->check(dm_jtm_shard_range((int64_t)n, (int)r, (int)engines.size(), &lo, &hi));
      engines[r]->check(dm_jtm_cache_rows_range(engines[r]->handle(), rowOff_.data(), rowIds_.data(), (int64_t)n, L_, lo, hi));
      hs.push_back(engines[r]->handle());
    }
    struct Drop { const std::vector<Engine *> &es; int L; ~Drop() { for (Engine *e : es) dm_jtm_cache_rows(e->handle(), nullptr, nullptr, 0, L); } } drop{engines, L_};
    engines.at(0)->check(dm_jtm_optimize_all(hs.data(), (int)hs.size(), itemCode_.data(), (int64_t)n, maxLevel_, gap_, hier_ ? 1 : 0, minLevel_,
                                             useMask_ ? 1 : 0, proj.data(), nullptr));
    std::map<int32_t, int32_t> res;
    for (size_t i = 0; i < n; i++) res[items_[i]] = proj[i];
    return res;
  }

 private:
  Engine &e_;
  int maxLevel_, gap_, L_;
  bool hier_;
  int minLevel_;
  bool useMask_;
  std::vector<int32_t> items_, itemCode_, rowIds_;
  std::vector<int64_t> rowOff_;
};

}  // namespace dm
