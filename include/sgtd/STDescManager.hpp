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
#include "../../adapter/STDesc_shim.hpp"

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
  double &operator()(int r, int c) { return m[r][c]; }
  const double &operator()(int r, int c) const { return m[r][c]; }
};

struct LOOP_RESULT {                       // STDesc.h:99-105
  int match_id = -1;
  int match_fitness = -1;                  // an int in the reference: verify_score is truncated into it (quirk 15)
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
  std::vector<int> device_ids_;   // several ids: the table sharded over these GPUs behind the one manager
};

class STDescManager {
 public:
  ConfigSetting config_setting_;
  int CS1 = 0;                         // whole ms of the last candidate_selector (STDesc.h:348)
  unsigned int current_frame_id_ = 0;  // STDesc.h:350

  explicit STDescManager(const ConfigSetting &cfg) : config_setting_(cfg) {
    status_ = cfg.device_ids_.empty() ? sgtd_shim::create(cfg, cfg.max_frame_n_, &h_, cfg.device_id_)
                                      : sgtd_shim::create(cfg, cfg.max_frame_n_, &h_, cfg.device_ids_);
    if (status_ != SGTD_OK)   // the reference's constructor cannot fail; a missing device is fatal here
      throw std::runtime_error(std::string("sgtd_create: ") + sgtd_strerror(status_));
  }
  ~STDescManager() { sgtd_shim::release_thread_buffers(); sgtd_destroy(h_); }   // (the page-locked fetch buffers of the calling thread go first)
  STDescManager(const STDescManager &) = delete;
  STDescManager &operator=(const STDescManager &) = delete;

  int last_status() const { return status_; }
  sgtd_handle handle() const { return h_; }

  // the bodies are the adapter's (adapter/STDesc_shim.hpp), instantiated with the plain structs above
  // STDesc.cpp:174-315
  void BuildSingleScanSTD(const std::vector<PointXYZL> &instance_pc, std::vector<STDesc> &stds_vec) {
    status_ = sgtd_shim::BuildSingleScanSTD(h_, instance_pc, stds_vec);
  }

  // STDesc.cpp:149-172
  void AddSTDescs(const std::vector<STDesc> &stds_vec) { status_ = sgtd_shim::AddSTDescs(h_, stds_vec, current_frame_id_); }

  // STDesc.cpp:318-460
  void candidate_selector(const std::vector<STDesc> &stds_vec, std::vector<STDMatchList> &candidate_matcher_vec) {
    status_ = sgtd_shim::candidate_selector(h_, stds_vec, candidate_matcher_vec, current_frame_id_,
                                            config_setting_.candidate_num_, CS1);
  }

  // STDesc.cpp:84-147: candidate_selector, then candidate_verify (:462-547, on the device:
  // sgtd_verify) for every candidate, best strictly-largest score above icp_threshold_
  void SearchLoop(const std::vector<STDesc> &stds_vec, std::pair<int, double> &loop_result,
                  std::pair<Vec3, Mat3> &loop_transform,
                  std::vector<std::pair<STDesc, STDesc>> &loop_std_pair,
                  std::vector<LOOP_RESULT> &match_result_list) {
    status_ = sgtd_shim::SearchLoop(h_, stds_vec, loop_result, loop_transform, loop_std_pair, match_result_list,
                                    current_frame_id_, config_setting_.candidate_num_, config_setting_.icp_threshold_, CS1);
  }

 private:
  sgtd_handle h_ = nullptr;
  int status_ = SGTD_OK;
};

}  // namespace sgtd
