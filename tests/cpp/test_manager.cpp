// C++ host-side mirror (include/sgtd/STDescManager.hpp) against the CPU oracle,
// written the way a test of the reference's own STDescManager would read.
// Built and run by tests/test_cpp_host.py on the GPU box.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/sgtd/STDescManager.hpp"
#include "../../oracle/sgtd_oracle.h"

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                     \
    }                                                                   \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {   // xorshift64*
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv) {
  const int n_frames = 12, n_kp = 64;
  // landmarks on a 90 m square, frames = noisy re-observations from a slowly moving window
  std::vector<sgtd::PointXYZL> land(400);
  for (auto &p : land) {
    p.x = (float)(urand() * 90.0); p.y = (float)(urand() * 90.0); p.z = (float)(urand() * 3.0);
    p.label = 3 + (uint32_t)(urand() * 9.0);
  }
  auto frame = [&](int f, double noise) {
    std::vector<sgtd::PointXYZL> pc;
    const double cx = 20.0 + 4.0 * f, cy = 45.0;
    std::vector<std::pair<double, int>> order;
    for (int i = 0; i < (int)land.size(); i++)
      order.push_back({std::hypot(land[i].x - cx, land[i].y - cy), i});
    std::sort(order.begin(), order.end());
    for (int k = 0; k < n_kp; k++) {
      sgtd::PointXYZL p = land[order[k].second];
      p.x = (float)(p.x - cx + noise * (urand() - 0.5)); p.y = (float)(p.y - cy + noise * (urand() - 0.5));
      p.z = (float)(p.z + noise * (urand() - 0.5));
      pc.push_back(p);
    }
    return pc;
  };

  sgtd::ConfigSetting cfg;
  if (argc > 1 && std::string(argv[1]) == "multi") cfg.device_ids_ = {0, 0};   // the table sharded over two "devices" behind the one manager
  sgtd::STDescManager *std_manager = new sgtd::STDescManager(cfg);   // as semantic_graph_localization.cpp:417
  orc_config oc{cfg.descriptor_near_num_, cfg.candidate_num_, cfg.max_frame_n_, 1, cfg.descriptor_min_len_,
                cfg.descriptor_max_len_, cfg.std_side_resolution_, cfg.rough_dis_threshold_};
  orc_manager *oracle = orc_create(&oc);

  auto to_arrays = [](const std::vector<sgtd::PointXYZL> &pc, std::vector<float> &xyz, std::vector<uint32_t> &lab) {
    xyz.clear(); lab.clear();
    for (auto &p : pc) { xyz.push_back(p.x); xyz.push_back(p.y); xyz.push_back(p.z); lab.push_back(p.label); }
  };

  std::vector<float> xyz; std::vector<uint32_t> lab;
  for (int f = 0; f < n_frames; f++) {                       // the map loop, :419-458
    auto map_cloud = frame(f, 0.02);
    std::vector<sgtd::STDesc> map_stds_vec;
    std_manager->BuildSingleScanSTD(map_cloud, map_stds_vec);
    CHECK(std_manager->last_status() == SGTD_OK);
    to_arrays(map_cloud, xyz, lab);
    const int64_t n_ref = orc_build(oracle, xyz.data(), lab.data(), n_kp);
    CHECK((int64_t)map_stds_vec.size() == n_ref);
    for (auto &d : map_stds_vec) CHECK(d.frame_id_ == (unsigned)f);
    std_manager->AddSTDescs(map_stds_vec);
    orc_add_last(oracle);
    CHECK(std_manager->current_frame_id_ == (unsigned)(f + 1));
    CHECK(orc_current_frame_id(oracle) == (uint32_t)(f + 1));
  }

  for (int qf : {3, 8}) {                                   // the query loop, :567-604
    auto query_cloud = frame(qf, 0.08);
    std::vector<sgtd::STDesc> query_stds_vec;
    std_manager->BuildSingleScanSTD(query_cloud, query_stds_vec);
    std::vector<sgtd::STDMatchList> candidate_matcher_vec;
    std_manager->candidate_selector(query_stds_vec, candidate_matcher_vec);
    CHECK(std_manager->last_status() == SGTD_OK);

    to_arrays(query_cloud, xyz, lab);
    orc_build(oracle, xyz.data(), lab.data(), n_kp);
    std::vector<int32_t> cf(cfg.candidate_num_), cv(cfg.candidate_num_);
    std::vector<int64_t> co(cfg.candidate_num_ + 1);
    int32_t nc = 0;
    orc_select(oracle, 1, nullptr, 0, cf.data(), cv.data(), co.data(), &nc);
    CHECK((int)candidate_matcher_vec.size() == nc);
    CHECK(nc > 0);
    std::vector<int32_t> qi(orc_cand_match_total(oracle));
    std::vector<int64_t> en(qi.size());
    orc_cand_matches(oracle, qi.data(), en.data());
    for (int k = 0; k < nc; k++) {
      const auto &ml = candidate_matcher_vec[k];
      CHECK(ml.match_id_.first == n_frames && ml.match_id_.second == cf[k]);
      CHECK((int64_t)ml.match_list_.size() == co[k + 1] - co[k]);
      CHECK((int)ml.match_list_.size() == cv[k]);
      // pair.second must be the table entry the oracle names: compare its geometry
      std::vector<double> side(3), vertex(9);
      std::vector<int32_t> label(3); uint32_t fr = 0;
      orc_desc_soa out{side.data(), nullptr, nullptr, vertex.data(), label.data(), &fr, nullptr};
      for (size_t r = 0; r < ml.match_list_.size(); r += 7) {
        const int64_t e = en[co[k] + r];
        orc_fetch_entries(oracle, &e, 1, &out);
        const sgtd::STDesc &db = ml.match_list_[r].second;
        CHECK(db.frame_id_ == fr && (int)fr == cf[k]);
        for (int c = 0; c < 3; c++) {
          CHECK(db.side_length_[c] == side[c]);
          CHECK(db.vertex_A_[c] == vertex[c] && db.vertex_B_[c] == vertex[3 + c] && db.vertex_C_[c] == vertex[6 + c]);
          CHECK((int)db.vertex_attached_[c] == label[c]);
        }
        const sgtd::STDesc &qd = ml.match_list_[r].first;
        CHECK(qd.side_length_[0] == query_stds_vec[qi[co[k] + r]].side_length_[0]);
      }
    }
    std::printf("query of frame %d: %d candidates, top-1 frame %d with %d votes\n", qf, nc, cf[0], cv[0]);
    CHECK(std::abs(cf[0] - qf) <= 2);

    // SearchLoop (STDesc.cpp:84-147) against candidate_verify of the restatement
    std::pair<int, double> loop_result;
    std::pair<sgtd::Vec3, sgtd::Mat3> loop_transform;
    std::vector<std::pair<sgtd::STDesc, sgtd::STDesc>> loop_std_pair;
    std::vector<sgtd::LOOP_RESULT> match_result_list;
    std_manager->SearchLoop(query_stds_vec, loop_result, loop_transform, loop_std_pair, match_result_list);
    CHECK(std_manager->last_status() == SGTD_OK);
    CHECK((int)match_result_list.size() == nc);
    double best_score = 0; int best = -1;
    for (int k = 0; k < nc; k++) {
      double t[3], rot[9];
      std::vector<int32_t> sidx(co[k + 1] - co[k] + 1);
      int32_t ns = 0;
      const double sc = orc_verify(oracle, k, t, rot, sidx.data(), &ns);
      CHECK(match_result_list[k].match_fitness == sc);
      CHECK(match_result_list[k].match_id == cf[k]);
      if (sc >= 0) {
        CHECK((int)match_result_list[k].loop_std_pair.size() == ns);
        // sucess_match_vec: the pairs the restatement names (positions in candidate k's match list), both sides
        std::vector<double> side(3), vertex(9);
        std::vector<int32_t> label(3), node(3); uint32_t fr = 0;
        orc_desc_soa out{side.data(), nullptr, nullptr, vertex.data(), label.data(), &fr, node.data()};
        for (int r = 0; r < ns; r += 3) {
          const std::pair<sgtd::STDesc, sgtd::STDesc> &pr = match_result_list[k].loop_std_pair[r];
          const int64_t at = co[k] + sidx[r], e = en[at];
          orc_fetch_entries(oracle, &e, 1, &out);
          CHECK(pr.second.frame_id_ == fr && (int)fr == cf[k]);
          const sgtd::STDesc &qd = query_stds_vec[qi[at]];
          for (int c = 0; c < 3; c++) {
            CHECK(pr.second.side_length_[c] == side[c]);
            CHECK(pr.second.vertex_A_[c] == vertex[c] && pr.second.vertex_B_[c] == vertex[3 + c] && pr.second.vertex_C_[c] == vertex[6 + c]);
            CHECK((int)pr.second.vertex_attached_[c] == label[c]);
            CHECK(pr.second.node_id.size() == 3 && pr.second.node_id[c] == node[c]);
            CHECK(pr.first.side_length_[c] == qd.side_length_[c] && pr.first.vertex_A_[c] == qd.vertex_A_[c]);
            CHECK(pr.first.node_id.size() == 3 && pr.first.node_id[c] == qd.node_id[c]);
          }
        }
        for (int a = 0; a < 3; a++) {
          CHECK(match_result_list[k].loop_transform.first[a] == t[a]);
          for (int b = 0; b < 3; b++) CHECK(match_result_list[k].loop_transform.second.m[a][b] == rot[a * 3 + b]);
        }
      }
      if (sc > best_score) { best_score = sc; best = k; }
    }
    if (best_score > cfg.icp_threshold_) {
      CHECK(loop_result.first == cf[best] && loop_result.second == best_score);
      CHECK((int)loop_std_pair.size() == (int)best_score);
    } else {
      CHECK(loop_result.first == -1 && loop_result.second == 0);
    }
    std::printf("SearchLoop: frame %d score %.0f\n", loop_result.first, loop_result.second);
  }
  orc_destroy(oracle);
  delete std_manager;
  std::printf("cpp host mirror ok\n");
  return 0;
}
