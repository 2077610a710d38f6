// Exercises the C++ host facade (include/dismember.hpp) on the GPU; driven by tests/test_gpu_cpp_facade.py, which
// writes the inputs as raw arrays into a directory and compares this program's JSON output with the Python facade /
// the CPU oracle.  usage: facade_test <dir>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "dismember.hpp"

template <typename T>
static std::vector<T> rd(const std::string &dir, const std::string &name) {
  std::ifstream f(dir + "/" + name, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("missing " + name);
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<T> v((size_t)n / sizeof(T));
  f.read((char *)v.data(), n);
  return v;
}

static void printRecs(const char *key, const dm::Recs &r, bool last = false) {
  std::printf("\"%s\": [", key);
  for (size_t i = 0; i < r.size(); i++) std::printf("%s[%d, %.17g]", i ? ", " : "", r[i].first, r[i].second);
  std::printf("]%s\n", last ? "" : ",");
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string d = argv[1];
  try {
    const auto meta = rd<int32_t>(d, "meta.i32");       // max_level, E, topk, beam, L
    const int maxLevel = meta[0], E = meta[1], topk = meta[2], beam = meta[3], L = meta[4];
    dm::Engine eng(0);
    eng.loadTree(rd<int32_t>(d, "codes.i32"), rd<int32_t>(d, "ids.i32"), rd<uint8_t>(d, "is_leaf.u8"), maxLevel,
                 rd<int32_t>(d, "leaf_ids.i32"), rd<int32_t>(d, "leaf_codes.i32"));
    eng.loadWeightsDin(rd<float>(d, "w32.f32"), E, ((int64_t)1 << (maxLevel + 1)) - 1);
    const auto query = rd<int32_t>(d, "query.i32");
    const auto consumed = rd<int32_t>(d, "consumed.i32");
    dm::TDM tdm(eng, "DIN");
    std::printf("{\n");
    printRecs("tdm_recommend", tdm.recommend(query, topk, beam));
    {   // cloneModule(): a worker's engine reads the owner's model (dm_clone) and recommends the same
      auto worker = eng.cloneEngine();
      dm::TDM tdm2(*worker, "DIN");
      printRecs("tdm_recommend_clone", tdm2.recommend(query, topk, beam));
    }
    {
      const auto a = tdm.recommendItems(query, topk, beam);
      const auto b = tdm.recommendItems(query, topk, beam, &consumed);
      std::printf("\"items_plain\": [");
      for (size_t i = 0; i < a.size(); i++) std::printf("%s%d", i ? ", " : "", a[i]);
      std::printf("],\n\"items_consumed\": [");
      for (size_t i = 0; i < b.size(); i++) std::printf("%s%d", i ? ", " : "", b[i]);
      std::printf("],\n");
    }
    {
      const auto seqs = rd<int32_t>(d, "batch.i32");
      const auto r = tdm.recommendBatch(seqs, (int64_t)seqs.size() / L, L, topk, beam);
      std::printf("\"batch_first_ids\": [");
      for (size_t u = 0; u < r.size(); u++) std::printf("%s%d", u ? ", " : "", r[u].empty() ? -1 : r[u][0].first);
      std::printf("],\n");
    }
    {   // metrics
      double p, r, n;
      dm::Metrics::computeMetrics({7, 1, 9, 4}, {9, 7, 100}, p, r, n);
      std::printf("\"metrics\": [%.17g, %.17g, %.17g],\n", p, r, n);
    }
    {   // JTM on the same tree: rows per item
      const auto rowItems = rd<int32_t>(d, "jtm_row_items.i32");     // item id per row
      const auto rows = rd<int32_t>(d, "jtm_rows.i32");              // [n_rows x L]
      std::map<int32_t, std::vector<int32_t>> itemRows;
      for (size_t k = 0; k < rowItems.size(); k++)
        itemRows[rowItems[k]].insert(itemRows[rowItems[k]].end(), rows.begin() + (ptrdiff_t)(k * (size_t)L), rows.begin() + (ptrdiff_t)((k + 1) * (size_t)L));
      dm::JTM jtm(eng, rd<int32_t>(d, "leaf_ids.i32"), rd<int32_t>(d, "leaf_codes.i32"), maxLevel, itemRows, 2, L);
      const auto proj = jtm.optimize();
      {   // the multi-worker entry with a single worker (dm_jtm_optimize_all, n == 1) and a one-rank RCCL communicator attached
        const auto again = jtm.optimizeAll({&eng});
        dm_comm_t c1 = nullptr;
        char id[DM_COMM_ID_BYTES];
        if (dm_comm_unique_id(id) != DM_OK || dm_comm_create_rccl(1, 0, id, 0, &c1) != DM_OK) throw dm::Error(DM_ERR_HIP, "one-rank communicator");
        eng.attachComm(c1);
        const auto withComm = jtm.optimize();
        eng.attachComm(nullptr);
        dm_comm_destroy(c1);
        std::printf("\"jtm_all_equal\": %s,\n", (again == proj && withComm == proj) ? "true" : "false");
      }
      std::printf("\"jtm_projection\": [");
      bool first = true;
      for (auto &kv : proj) { std::printf("%s[%d, %d]", first ? "" : ", ", kv.first, kv.second); first = false; }
      std::printf("],\n");
    }
    {   // OTM: f64 weights, item -> node mapping
      dm::Engine e2(0);
      const auto w64 = rd<double>(d, "w64.f64");
      e2.loadWeightsDin(w64, E, ((int64_t)1 << (maxLevel + 1)) - 1);
      const auto m = rd<int32_t>(d, "otm_mapping.i32");              // pairs (item, node)
      std::map<int32_t, int32_t> mp;
      for (size_t i = 0; i + 1 < m.size(); i += 2) mp[m[i]] = m[i + 1];
      dm::OTM otm(e2, mp);
      printRecs("otm_recommend", otm.recommend(rd<int32_t>(d, "otm_query.i32"), topk, beam));
    }
    {   // Deep-Retrieval: model + path table through the C ABI, recommend through the facade
      const auto dm_ = rd<int32_t>(d, "dr_meta.i32");                // E, L, K, D, num_item, beam, topk
      const int dE = dm_[0], dL = dm_[1], dK = dm_[2], dD = dm_[3], dn = dm_[4];
      dm::Engine e3(0);
      const auto emb = rd<double>(d, "dr_layer_emb.f64");
      std::vector<std::vector<double>> W, B;
      std::vector<const void *> wp, bp;
      for (int i = 0; i < dD; i++) {
        W.push_back(rd<double>(d, "dr_w" + std::to_string(i) + ".f64"));
        B.push_back(rd<double>(d, "dr_b" + std::to_string(i) + ".f64"));
      }
      for (int i = 0; i < dD; i++) { wp.push_back(W[(size_t)i].data()); bp.push_back(B[(size_t)i].data()); }
      const auto remb = rd<double>(d, "dr_rerank_emb.f64"), rw = rd<double>(d, "dr_rerank_w.f64"), rb = rd<double>(d, "dr_rerank_b.f64");
      const auto sw = rd<double>(d, "dr_softmax_w.f64"), sb = rd<double>(d, "dr_softmax_b.f64");
      dm_dr_model m{};
      m.dtype = DM_F64; m.on_device = 0; m.embed = dE; m.seq_len = dL; m.num_node = dK; m.num_layer = dD; m.num_item = dn;
      m.layer_emb = emb.data(); m.layer_w = wp.data(); m.layer_b = bp.data();
      m.rerank_emb = remb.data(); m.rerank_w = rw.data(); m.rerank_b = rb.data(); m.softmax_w = sw.data(); m.softmax_b = sb.data();
      e3.check(dm_dr_load_model(e3.handle(), &m));
      const auto pn = rd<int32_t>(d, "dr_path_nodes.i32");
      const auto po = rd<int64_t>(d, "dr_item_off.i64");
      const auto pi = rd<int32_t>(d, "dr_items.i32");
      e3.check(dm_dr_load_path_items(e3.handle(), pn.data(), (int64_t)po.size() - 1, po.data(), pi.data()));
      std::map<int32_t, int32_t> idmap;
      for (int i = 0; i < dn; i++) idmap[1000 + 3 * i] = i;          // item = 1000 + 3 * id
      dm::DeepRetrieval dr(e3, idmap);
      printRecs("dr_recommend", dr.recommend(rd<int32_t>(d, "dr_query.i32"), dm_[6], dm_[5]));
    }
    {   // error mapping
      int code = 0;
      try { tdm.recommend(std::vector<int32_t>(40, 1), topk, beam); } catch (const dm::Error &e) { code = e.code; }
      std::printf("\"error_code_L40\": %d\n", code);
    }
    std::printf("}\n");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "facade_test: %s\n", e.what());
    return 1;
  }
  return 0;
}
