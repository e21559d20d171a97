// STDescManager.hpp — C++ host side above the C ABI (include/sgtd_accel.h).
//
// Mirrors the reference's operator API for the hot path with the same method
// names, argument meaning and error behaviour
// (src/sgtd/include/desc/STDesc.h:342-440):
//
//   void BuildSingleScanSTD(cloud, stds_vec)            STDesc.cpp:174-315
//   void AddSTDescs(stds_vec)                           STDesc.cpp:149-172
//   void candidate_selector(stds_vec, matcher_vec)      STDesc.cpp:318-460
//   public state: current_frame_id_, config_setting_, CS1
//
// Eigen/PCL/ROS are not available in this repository's build image, so the
// containers here are plain structs with the reference's field names
// (Vec3 for Eigen::Vector3d, PointXYZL for pcl::PointXYZL); INTEGRATION.md
// shows the Eigen/PCL flavoured adapter a maintainer of the reference would
// drop into STDesc.cpp.  No exceptions cross the C ABI; like the reference the
// methods return void, failures are reported through last_status().
#pragma once
#include <chrono>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../sgtd_accel.h"

namespace sgtd {

struct Vec3 {
  double v[3] = {0, 0, 0};
  double &operator[](int i) { return v[i]; }
  const double &operator[](int i) const { return v[i]; }
};

struct PointXYZL {   // pcl::PointXYZL fields the path reads (utility.hpp:646-659)
  float x, y, z;
  uint32_t label;
};

// STDesc (STDesc.h:75-97) without the never-read covariance matrices
struct STDesc {
  Vec3 side_length_;
  Vec3 angle_;
  Vec3 center_;
  unsigned int frame_id_ = 0;
  Vec3 vertex_A_, vertex_B_, vertex_C_;
  Vec3 vertex_attached_;
  std::vector<int> node_id;
};

// STDMatchList (STDesc.h:120-124)
struct STDMatchList {
  std::vector<std::pair<STDesc, STDesc>> match_list_;
  std::pair<int, int> match_id_;
  double mean_dis_ = 0;
};

// ConfigSetting fields of the path (STDesc.h:38-72); defaults = shipped YAML
struct Mat3 {
  double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
};

struct LOOP_RESULT {                       // STDesc.h:99-105
  int match_id = -1;
  double match_fitness = -1;               // the reference stores the score in an int (quirk 15)
  std::pair<Vec3, Mat3> loop_transform;
  std::vector<std::pair<STDesc, STDesc>> loop_std_pair;
};

struct ConfigSetting {
  int descriptor_near_num_ = 10;
  double descriptor_min_len_ = 0.5;
  double descriptor_max_len_ = 50;
  double std_side_resolution_ = 1;
  int candidate_num_ = 50;
  double rough_dis_threshold_ = 0.03;
  double icp_threshold_ = 0.5;          // STDesc.h:68 (SG_localization.yaml:89 ships 0.4)
  int max_frame_n_ = 20000;   // MAX_FRAME_N (STDesc.h:33)
  int device_id_ = 0;
};

class STDescManager {
 public:
  ConfigSetting config_setting_;
  int CS1 = 0;                         // whole ms of the last candidate_selector (STDesc.h:348)
  unsigned int current_frame_id_ = 0;  // STDesc.h:350

  explicit STDescManager(const ConfigSetting &cfg) : config_setting_(cfg) {
    sgtd_config c;
    sgtd_default_config(&c);
    c.descriptor_near_num = cfg.descriptor_near_num_;
    c.descriptor_min_len = cfg.descriptor_min_len_;
    c.descriptor_max_len = cfg.descriptor_max_len_;
    c.std_side_resolution = cfg.std_side_resolution_;
    c.candidate_num = cfg.candidate_num_;
    c.rough_dis_threshold = cfg.rough_dis_threshold_;
    c.max_frame_n = cfg.max_frame_n_;
    c.device_id = cfg.device_id_;
    status_ = sgtd_create(&c, &h_);
    if (status_ != SGTD_OK)   // the reference's constructor cannot fail; a missing device is fatal here
      throw std::runtime_error(std::string("sgtd_create: ") + sgtd_strerror(status_));
  }
  ~STDescManager() { sgtd_destroy(h_); }
  STDescManager(const STDescManager &) = delete;
  STDescManager &operator=(const STDescManager &) = delete;

  int last_status() const { return status_; }
  sgtd_handle handle() const { return h_; }

  // STDesc.cpp:174-315
  void BuildSingleScanSTD(const std::vector<PointXYZL> &instance_pc, std::vector<STDesc> &stds_vec) {
    stds_vec.clear();
    const int n = (int)instance_pc.size();
    std::vector<float> xyz(3 * (size_t)n);
    std::vector<uint32_t> label(n);
    for (int i = 0; i < n; i++) {
      xyz[3 * i] = instance_pc[i].x; xyz[3 * i + 1] = instance_pc[i].y; xyz[3 * i + 2] = instance_pc[i].z;
      label[i] = instance_pc[i].label;
    }
    Soa s((size_t)sgtd_max_descs(h_, n));
    int64_t n_out = 0;
    status_ = sgtd_build(h_, xyz.data(), label.data(), n, &s.view, (int64_t)s.cap, &n_out);
    if (status_ != SGTD_OK) return;
    s.to_descs(0, (size_t)n_out, stds_vec);
  }

  // STDesc.cpp:149-172
  void AddSTDescs(const std::vector<STDesc> &stds_vec) {
    Soa s(stds_vec.size());
    s.from_descs(stds_vec);
    status_ = sgtd_add(h_, &s.view, (int64_t)stds_vec.size());
    sgtd_current_frame_id(h_, &current_frame_id_);
  }

  // STDesc.cpp:318-460
  void candidate_selector(const std::vector<STDesc> &stds_vec, std::vector<STDMatchList> &candidate_matcher_vec) {
    auto t1 = std::chrono::high_resolution_clock::now();
    Soa q(stds_vec.size());
    q.from_descs(stds_vec);
    status_ = sgtd_query_descs(h_, &q.view, (int64_t)stds_vec.size());
    if (status_ != SGTD_OK) return;
    const int cn = config_setting_.candidate_num_;
    int32_t n_cand = 0;
    std::vector<int32_t> frame(cn), votes(cn);
    std::vector<int64_t> off(cn + 1);
    status_ = sgtd_result_candidates(h_, &n_cand, frame.data(), votes.data(), off.data());
    if (status_ != SGTD_OK) return;
    const int64_t total = off[n_cand];
    std::vector<int32_t> qi(total);
    std::vector<int64_t> en(total);
    int64_t got = 0;
    status_ = sgtd_result_pairs(h_, 0, qi.data(), en.data(), total, &got);
    if (status_ != SGTD_OK) return;
    Soa ent((size_t)total);
    status_ = sgtd_fetch_entries(h_, en.data(), total, &ent.view);
    if (status_ != SGTD_OK) return;
    std::vector<STDesc> db;
    ent.to_descs(0, (size_t)total, db);
    for (int k = 0; k < n_cand; k++) {
      STDMatchList ml;
      ml.match_id_.first = (int)current_frame_id_;   // :436
      ml.match_id_.second = frame[k];
      for (int64_t r = off[k]; r < off[k + 1]; r++) ml.match_list_.emplace_back(stds_vec[qi[r]], db[r]);
      candidate_matcher_vec.push_back(std::move(ml));
    }
    auto t2 = std::chrono::high_resolution_clock::now();
    CS1 = (int)(std::chrono::duration<double>(t2 - t1).count() * 1000);   // int truncation as :455
  }

  // STDesc.cpp:84-147: candidate_selector, then candidate_verify (:462-547, on the device:
  // sgtd_verify) for every candidate, best strictly-largest score above icp_threshold_
  void SearchLoop(const std::vector<STDesc> &stds_vec, std::pair<int, double> &loop_result,
                  std::pair<Vec3, Mat3> &loop_transform,
                  std::vector<std::pair<STDesc, STDesc>> &loop_std_pair,
                  std::vector<LOOP_RESULT> &match_result_list) {
    if (stds_vec.empty()) {                        // "No STDescs!" (:89-93)
      loop_result = std::pair<int, double>(-1, 0);
      return;
    }
    std::vector<STDMatchList> candidate_matcher_vec;
    candidate_selector(stds_vec, candidate_matcher_vec);
    if (status_ != SGTD_OK) { loop_result = std::pair<int, double>(-1, 0); return; }
    status_ = sgtd_verify(h_);
    if (status_ != SGTD_OK) { loop_result = std::pair<int, double>(-1, 0); return; }
    const int cn = config_setting_.candidate_num_;
    std::vector<double> score(cn), pose((size_t)cn * 12);
    status_ = sgtd_result_verify(h_, 0, score.data(), pose.data());
    if (status_ != SGTD_OK) { loop_result = std::pair<int, double>(-1, 0); return; }
    double best_score = 0;
    int best = -1;
    for (size_t i = 0; i < candidate_matcher_vec.size(); i++) {     // :105-131
      LOOP_RESULT r;
      r.match_id = candidate_matcher_vec[i].match_id_.second;
      r.match_fitness = score[i];
      for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) r.loop_transform.second.m[a][b] = pose[i * 12 + a * 3 + b];
        r.loop_transform.first[a] = pose[i * 12 + 9 + a];
      }
      if (score[i] >= 0) {
        const auto &ml = candidate_matcher_vec[i].match_list_;
        std::vector<int32_t> idx(ml.size());
        int64_t n = 0;
        status_ = sgtd_result_inliers(h_, 0, (int)i, idx.data(), (int64_t)idx.size(), &n);
        if (status_ != SGTD_OK) { loop_result = std::pair<int, double>(-1, 0); return; }
        for (int64_t k = 0; k < n; k++) r.loop_std_pair.push_back(ml[idx[k]]);
      }
      if (score[i] > best_score) { best_score = score[i]; best = (int)i; }
      match_result_list.push_back(std::move(r));
    }
    if (best_score > config_setting_.icp_threshold_) {             // :138-146
      const LOOP_RESULT &b = match_result_list[match_result_list.size() - candidate_matcher_vec.size() + best];
      loop_result = std::pair<int, double>(b.match_id, best_score);
      loop_transform = b.loop_transform;
      loop_std_pair = b.loop_std_pair;
    } else {
      loop_result = std::pair<int, double>(-1, 0);
    }
  }

 private:
  struct Soa {
    std::vector<double> side, angle, center;
    std::vector<float> vertex;
    std::vector<int32_t> label, node_id;
    std::vector<uint32_t> frame;
    sgtd_desc_soa view;
    size_t cap;
    explicit Soa(size_t n) : side(3 * n), angle(3 * n), center(3 * n), vertex(9 * n), label(3 * n),
                             node_id(3 * n), frame(n), cap(n) {
      view.side = side.data(); view.angle = angle.data(); view.center = center.data();
      view.vertex = vertex.data(); view.label = label.data(); view.frame = frame.data();
      view.node_id = node_id.data();
    }
    void from_descs(const std::vector<STDesc> &v) {
      for (size_t i = 0; i < v.size(); i++) {
        for (int k = 0; k < 3; k++) {
          side[3 * i + k] = v[i].side_length_[k]; angle[3 * i + k] = v[i].angle_[k];
          center[3 * i + k] = v[i].center_[k];
          vertex[9 * i + k] = (float)v[i].vertex_A_[k]; vertex[9 * i + 3 + k] = (float)v[i].vertex_B_[k];
          vertex[9 * i + 6 + k] = (float)v[i].vertex_C_[k];
          label[3 * i + k] = (int32_t)v[i].vertex_attached_[k];
          node_id[3 * i + k] = v[i].node_id.size() == 3 ? v[i].node_id[k] : 0;
        }
        frame[i] = v[i].frame_id_;
      }
    }
    void to_descs(size_t first, size_t n, std::vector<STDesc> &out) const {
      out.resize(n);
      for (size_t i = 0; i < n; i++) {
        const size_t s = first + i;
        STDesc &d = out[i];
        for (int k = 0; k < 3; k++) {
          d.side_length_[k] = side[3 * s + k]; d.angle_[k] = angle[3 * s + k]; d.center_[k] = center[3 * s + k];
          d.vertex_A_[k] = vertex[9 * s + k]; d.vertex_B_[k] = vertex[9 * s + 3 + k];
          d.vertex_C_[k] = vertex[9 * s + 6 + k];
          d.vertex_attached_[k] = (double)label[3 * s + k];
        }
        d.frame_id_ = frame[s];
        d.node_id = {node_id[3 * s], node_id[3 * s + 1], node_id[3 * s + 2]};
      }
    }
  };

  sgtd_handle h_ = nullptr;
  int status_ = SGTD_OK;
};

}  // namespace sgtd
