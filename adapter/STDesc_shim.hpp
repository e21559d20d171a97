// STDesc_shim.hpp — the adapter a maintainer of Hfx-J/SGTD drops into the reference so that
// the hot path of STDescManager runs on the MI355X library behind include/sgtd_accel.h.
//
// Reference interface kept unchanged (src/sgtd/include/desc/STDesc.h:342-440):
//
//   STDescManager::STDescManager(ConfigSetting &)                    STDesc.h:359-365
//   void BuildSingleScanSTD(const pcl::PointCloud<pcl::PointXYZL>::Ptr &, std::vector<STDesc> &)   STDesc.cpp:174-315
//   void AddSTDescs(const std::vector<STDesc> &)                                                    STDesc.cpp:149-172
//   void candidate_selector(const std::vector<STDesc> &, std::vector<STDMatchList> &)               STDesc.cpp:318-460
//   void SearchLoop(const std::vector<STDesc> &, std::pair<int, double> &,
//                   std::pair<Eigen::Vector3d, Eigen::Matrix3d> &,
//                   std::vector<std::pair<STDesc, STDesc>> &, std::vector<LOOP_RESULT> &)           STDesc.cpp:84-147
//
// How it is used in the reference tree (the whole patch):
//
//   // STDesc.h: one new member            sgtd_handle accel_ = nullptr;
//   // STDesc.cpp:
//   #include "STDesc_shim.hpp"
//   STDescManager::STDescManager(ConfigSetting &c) : config_setting_(c) {
//     current_frame_id_ = 0;
//     if (sgtd_shim::create(c, MAX_FRAME_N, &accel_) != SGTD_OK) ROS_FATAL_STREAM("sgtd_create failed");
//   }
//   void STDescManager::BuildSingleScanSTD(const pcl::PointCloud<pcl::PointXYZL>::Ptr &pc, std::vector<STDesc> &v) {
//     sgtd_shim::BuildSingleScanSTD(accel_, pc, v);
//   }
//   void STDescManager::AddSTDescs(const std::vector<STDesc> &v) { sgtd_shim::AddSTDescs(accel_, v, current_frame_id_); }
//   void STDescManager::candidate_selector(const std::vector<STDesc> &v, std::vector<STDMatchList> &m) {
//     sgtd_shim::candidate_selector(accel_, v, m, current_frame_id_, config_setting_.candidate_num_, CS1);
//   }
//   // optional (verification on the device as well; otherwise SearchLoop stays as it is):
//   void STDescManager::SearchLoop(...) { sgtd_shim::SearchLoop(accel_, stds_vec, loop_result, loop_transform,
//       loop_std_pair, match_result_list, current_frame_id_, config_setting_.candidate_num_,
//       config_setting_.icp_threshold_, CS1); }
//
// The functions are templates over the reference's own types (STDesc, STDMatchList, LOOP_RESULT,
// ConfigSetting, Eigen vectors, the PCL cloud pointer): nothing here includes Eigen, PCL or ROS,
// the types come from the translation unit that includes this header.  The same templates are
// instantiated with plain structs by include/sgtd/STDescManager.hpp (this repository's C++
// host, compiled and run by tests/cpp/test_manager.cpp on the GPU) and with look-alikes of the
// reference's Eigen/PCL-typed declarations by tests/cpp/test_shim.cpp (compile + link check).
// No exception crosses the C ABI; like the reference the methods return void — every function
// here returns the sgtd_status of the first failing call (SGTD_OK otherwise) for the caller to
// log.
#pragma once
#include <chrono>
#include <cstdint>
#include <utility>
#include <vector>

#ifndef SGTD_ACCEL_H
#include "sgtd_accel.h"   // include/sgtd_accel.h of this repository (add its directory to the include path)
#endif

namespace sgtd_shim {

// ---- point-cloud access: pcl::PointCloud<pcl::PointXYZL>::Ptr (->points) or a plain vector
template <class Ptr>
auto points_of(const Ptr &pc) -> decltype((pc->points)) { return pc->points; }
template <class P>
const std::vector<P> &points_of(const std::vector<P> &pc) { return pc; }

// ---- STDesc <-> sgtd_desc_soa ----------------------------------------------------------
struct SoaBuf {
  std::vector<double> side, angle, center;
  std::vector<float> vertex;
  std::vector<int32_t> label, node_id;
  std::vector<uint32_t> frame;
  sgtd_desc_soa v;
  explicit SoaBuf(size_t n) : side(3 * n), angle(3 * n), center(3 * n), vertex(9 * n), label(3 * n), node_id(3 * n), frame(n) {
    v.side = side.data(); v.angle = angle.data(); v.center = center.data(); v.vertex = vertex.data();
    v.label = label.data(); v.frame = frame.data(); v.node_id = node_id.data();
  }
  size_t capacity() const { return frame.size(); }
};

template <class Desc>
void to_soa(const std::vector<Desc> &in, SoaBuf &b) {
  for (size_t i = 0; i < in.size(); i++) {
    for (int k = 0; k < 3; k++) {
      b.side[3 * i + k] = in[i].side_length_[k]; b.angle[3 * i + k] = in[i].angle_[k]; b.center[3 * i + k] = in[i].center_[k];
      b.vertex[9 * i + k] = (float)in[i].vertex_A_[k]; b.vertex[9 * i + 3 + k] = (float)in[i].vertex_B_[k];
      b.vertex[9 * i + 6 + k] = (float)in[i].vertex_C_[k];
      b.label[3 * i + k] = (int32_t)in[i].vertex_attached_[k];          // (int) as STDesc.cpp:158-160
      b.node_id[3 * i + k] = in[i].node_id.size() == 3 ? in[i].node_id[k] : 0;
    }
    b.frame[i] = in[i].frame_id_;
  }
}

template <class Desc>
void from_soa(const SoaBuf &b, size_t n, std::vector<Desc> &out) {
  out.resize(n);
  for (size_t i = 0; i < n; i++) {
    Desc &d = out[i];
    for (int k = 0; k < 3; k++) {
      d.side_length_[k] = b.side[3 * i + k]; d.angle_[k] = b.angle[3 * i + k]; d.center_[k] = b.center[3 * i + k];
      d.vertex_A_[k] = b.vertex[9 * i + k]; d.vertex_B_[k] = b.vertex[9 * i + 3 + k]; d.vertex_C_[k] = b.vertex[9 * i + 6 + k];
      d.vertex_attached_[k] = (double)b.label[3 * i + k];
    }
    d.frame_id_ = b.frame[i];
    d.node_id = {b.node_id[3 * i], b.node_id[3 * i + 1], b.node_id[3 * i + 2]};
  }
}

// ---- constructor body (STDesc.h:359-365): ConfigSetting -> sgtd_config -------------------
// device_ids: the GPUs of this node the table is sharded over (one id = an ordinary handle)
template <class Config>
int create(const Config &cs, int max_frame_n, sgtd_handle *out, const std::vector<int> &device_ids) {
  sgtd_config c;
  sgtd_default_config(&c);
  c.descriptor_near_num = cs.descriptor_near_num_;
  c.descriptor_min_len = cs.descriptor_min_len_;
  c.descriptor_max_len = cs.descriptor_max_len_;
  c.std_side_resolution = cs.std_side_resolution_;
  c.candidate_num = cs.candidate_num_;
  c.rough_dis_threshold = cs.rough_dis_threshold_;
  c.max_frame_n = max_frame_n;
  return sgtd_create_multi(&c, device_ids.data(), (int)device_ids.size(), out);
}

template <class Config>
int create(const Config &cs, int max_frame_n, sgtd_handle *out, int device_id = 0) {
  sgtd_config c;
  sgtd_default_config(&c);
  c.descriptor_near_num = cs.descriptor_near_num_;
  c.descriptor_min_len = cs.descriptor_min_len_;
  c.descriptor_max_len = cs.descriptor_max_len_;
  c.std_side_resolution = cs.std_side_resolution_;
  c.candidate_num = cs.candidate_num_;
  c.rough_dis_threshold = cs.rough_dis_threshold_;
  c.max_frame_n = max_frame_n;
  c.device_id = device_id;
  return sgtd_create(&c, out);
}

// ---- STDesc.cpp:174-315 --------------------------------------------------------------------
template <class Cloud, class Desc>
int BuildSingleScanSTD(sgtd_handle h, const Cloud &instance_pc, std::vector<Desc> &stds_vec) {
  stds_vec.clear();
  const auto &pts = points_of(instance_pc);
  const int n = (int)pts.size();
  std::vector<float> xyz(3 * (size_t)n);
  std::vector<uint32_t> label((size_t)n);
  for (int i = 0; i < n; i++) {               // Graph2CloudL's fields, utility.hpp:646-659
    xyz[3 * i] = pts[i].x; xyz[3 * i + 1] = pts[i].y; xyz[3 * i + 2] = pts[i].z;
    label[i] = pts[i].label;
  }
  SoaBuf b((size_t)sgtd_max_descs(h, n));
  int64_t n_out = 0;
  const int st = sgtd_build(h, xyz.data(), label.data(), n, &b.v, (int64_t)b.capacity(), &n_out);
  if (st != SGTD_OK) return st;
  from_soa(b, (size_t)n_out, stds_vec);
  return SGTD_OK;
}

// ---- STDesc.cpp:149-172 --------------------------------------------------------------------
template <class Desc>
int AddSTDescs(sgtd_handle h, const std::vector<Desc> &stds_vec, unsigned int &current_frame_id) {
  SoaBuf b(stds_vec.size());
  to_soa(stds_vec, b);
  const int st = sgtd_add(h, &b.v, (int64_t)stds_vec.size());   // increments the frame counter first, like :151
  uint32_t id = current_frame_id;
  sgtd_current_frame_id(h, &id);
  current_frame_id = id;
  return st;
}

// ---- STDesc.cpp:318-460 --------------------------------------------------------------------
template <class Desc, class MatchList>
int candidate_selector(sgtd_handle h, const std::vector<Desc> &stds_vec, std::vector<MatchList> &candidate_matcher_vec,
                       unsigned int current_frame_id, int candidate_num, int &CS1) {
  const auto t1 = std::chrono::high_resolution_clock::now();
  SoaBuf q(stds_vec.size());
  to_soa(stds_vec, q);
  int st = sgtd_query_descs(h, &q.v, (int64_t)stds_vec.size());
  if (st != SGTD_OK) return st;
  const int cn = candidate_num;
  int32_t n_cand = 0;
  std::vector<int32_t> frame(cn), votes(cn);
  std::vector<int64_t> off(cn + 1);
  st = sgtd_result_candidates(h, &n_cand, frame.data(), votes.data(), off.data());
  if (st != SGTD_OK) return st;
  const int64_t total = off[n_cand];
  std::vector<int32_t> qi((size_t)total);
  std::vector<int64_t> en((size_t)total);
  int64_t got = 0;
  st = sgtd_result_pairs(h, 0, qi.data(), en.data(), total, &got);
  if (st != SGTD_OK) return st;
  SoaBuf ent((size_t)total);
  st = sgtd_fetch_entries(h, en.data(), total, &ent.v);
  if (st != SGTD_OK) return st;
  std::vector<Desc> db;
  from_soa(ent, (size_t)total, db);
  for (int k = 0; k < n_cand; k++) {
    MatchList ml;
    ml.match_id_.first = (int)current_frame_id;    // :436
    ml.match_id_.second = frame[k];                // :437
    for (int64_t r = off[k]; r < off[k + 1]; r++) ml.match_list_.emplace_back(stds_vec[qi[(size_t)r]], db[(size_t)r]);
    candidate_matcher_vec.push_back(std::move(ml));
  }
  const auto t2 = std::chrono::high_resolution_clock::now();
  CS1 = (int)(std::chrono::duration<double>(t2 - t1).count() * 1000);   // int CS1 truncates a double ms value, :455
  return SGTD_OK;
}

// ---- STDesc.cpp:84-147 with candidate_verify (:462-547) on the device ---------------------
// Vec3 / Mat3 = Eigen::Vector3d / Eigen::Matrix3d (operator[] and operator()(row, col))
template <class Desc, class Vec3, class Mat3, class LoopResult>
int SearchLoop(sgtd_handle h, const std::vector<Desc> &stds_vec, std::pair<int, double> &loop_result,
               std::pair<Vec3, Mat3> &loop_transform, std::vector<std::pair<Desc, Desc>> &loop_std_pair,
               std::vector<LoopResult> &match_result_list, unsigned int current_frame_id, int candidate_num,
               double icp_threshold, int &CS1) {
  struct ML {   // STDMatchList's two fields the loop needs
    std::vector<std::pair<Desc, Desc>> match_list_;
    std::pair<int, int> match_id_;
  };
  loop_result = std::pair<int, double>(-1, 0);
  if (stds_vec.empty()) return SGTD_OK;            // "No STDescs!" (:89-93)
  std::vector<ML> cands;
  int st = candidate_selector(h, stds_vec, cands, current_frame_id, candidate_num, CS1);
  if (st != SGTD_OK) return st;
  st = sgtd_verify(h);                             // :105-118 for every candidate
  if (st != SGTD_OK) return st;
  const int cn = candidate_num;
  std::vector<double> score(cn), pose((size_t)cn * 12);   // rot row-major (9), then t (3)
  st = sgtd_result_verify(h, 0, score.data(), pose.data());
  if (st != SGTD_OK) return st;
  double best_score = 0;
  int best = -1;
  const size_t first = match_result_list.size();
  for (size_t i = 0; i < cands.size(); i++) {      // :105-131
    LoopResult r;
    r.match_id = cands[i].match_id_.second;
    r.match_fitness = score[i];                    // an int member in the reference: truncates like :119
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) r.loop_transform.second(a, b) = pose[i * 12 + a * 3 + b];
      r.loop_transform.first[a] = pose[i * 12 + 9 + a];
    }
    if (score[i] >= 0) {                           // sucess_match_vec, :516-539
      const auto &ml = cands[i].match_list_;
      std::vector<int32_t> idx(ml.size());
      int64_t n = 0;
      st = sgtd_result_inliers(h, 0, (int)i, idx.data(), (int64_t)idx.size(), &n);
      if (st != SGTD_OK) return st;
      for (int64_t k = 0; k < n; k++) r.loop_std_pair.push_back(ml[(size_t)idx[(size_t)k]]);
    }
    if (score[i] > best_score) { best_score = score[i]; best = (int)i; }   // :125-131
    match_result_list.push_back(std::move(r));
  }
  if (best >= 0 && best_score > icp_threshold) {   // :138-146
    const LoopResult &b = match_result_list[first + (size_t)best];
    loop_result = std::pair<int, double>(b.match_id, best_score);
    loop_transform = b.loop_transform;
    loop_std_pair = b.loop_std_pair;
  }
  return SGTD_OK;
}

}  // namespace sgtd_shim
