// Compile + link + smoke check of adapter/STDesc_shim.hpp against declarations shaped like the
// reference's (src/sgtd/include/desc/STDesc.h:38-124,342-440): Eigen-style vectors with
// operator[], a matrix with operator()(row, col), a PCL-style cloud pointer with ->points, the
// reference's struct and field NAMES, and a class STDescManager whose methods have the
// reference's exact signatures and forward to the adapter.  Eigen, PCL and ROS are not in this
// image: the few look-alike types below exist only so that the adapter's templates are
// instantiated with the type SHAPES they meet in the reference tree.
#include <cstdio>
#include <memory>
#include <utility>
#include <vector>

namespace Eigen {   // look-alikes: just the members the adapter touches
struct Vector3d {
  double v[3] = {0, 0, 0};
  double &operator[](int i) { return v[i]; }
  const double &operator[](int i) const { return v[i]; }
};
struct Matrix3d {
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double &operator()(int r, int c) { return m[r * 3 + c]; }
  const double &operator()(int r, int c) const { return m[r * 3 + c]; }
};
}  // namespace Eigen
namespace pcl {
struct PointXYZL { float x, y, z; unsigned int label; };
template <class P> struct PointCloud {
  std::vector<P> points;
  typedef std::shared_ptr<PointCloud<P>> Ptr;
  size_t size() const { return points.size(); }
};
}  // namespace pcl

#define MAX_FRAME_N 20000   // STDesc.h:33

typedef struct ConfigSetting {   // field names of STDesc.h:38-72 that the path reads
  int descriptor_near_num_ = 10;
  double descriptor_min_len_ = 0.5;
  double descriptor_max_len_ = 50;
  double std_side_resolution_ = 1;
  int candidate_num_ = 50;
  double rough_dis_threshold_ = 0.03;
  double icp_threshold_ = 0.4;
} ConfigSetting;

typedef struct STDesc {          // STDesc.h:75-97
  Eigen::Vector3d side_length_;
  Eigen::Vector3d angle_;
  Eigen::Vector3d center_;
  unsigned int frame_id_;
  Eigen::Vector3d vertex_A_;
  Eigen::Vector3d vertex_B_;
  Eigen::Vector3d vertex_C_;
  Eigen::Vector3d vertex_attached_;
  std::vector<int> node_id;
  Eigen::Matrix3d cov_mat_A_;
  Eigen::Matrix3d cov_mat_B_;
  Eigen::Matrix3d cov_mat_C_;
} STDesc;

struct LOOP_RESULT {             // STDesc.h:99-105
  int match_id;
  int match_fitness;
  std::pair<Eigen::Vector3d, Eigen::Matrix3d> loop_transform;
  std::vector<std::pair<STDesc, STDesc>> loop_std_pair;
};

typedef struct STDMatchList {    // STDesc.h:120-124
  std::vector<std::pair<STDesc, STDesc>> match_list_;
  std::pair<int, int> match_id_;
  double mean_dis_;
} STDMatchList;

#include "../../adapter/STDesc_shim.hpp"

class STDescManager {            // the patched class: reference signatures, STDesc.h:342-440
 public:
  ConfigSetting config_setting_;
  int CS1 = 0;
  unsigned int current_frame_id_;
  sgtd_handle accel_ = nullptr;
  int status = SGTD_OK;
  explicit STDescManager(ConfigSetting &config_setting) : config_setting_(config_setting) {
    current_frame_id_ = 0;
    status = sgtd_shim::create(config_setting_, MAX_FRAME_N, &accel_);
  }
  ~STDescManager() { sgtd_destroy(accel_); }
  void BuildSingleScanSTD(const pcl::PointCloud<pcl::PointXYZL>::Ptr &instance_pc, std::vector<STDesc> &stds_vec) {
    status = sgtd_shim::BuildSingleScanSTD(accel_, instance_pc, stds_vec);
  }
  void AddSTDescs(const std::vector<STDesc> &stds_vec) { status = sgtd_shim::AddSTDescs(accel_, stds_vec, current_frame_id_); }
  void SearchLoop(const std::vector<STDesc> &stds_vec, std::pair<int, double> &loop_result,
                  std::pair<Eigen::Vector3d, Eigen::Matrix3d> &loop_transform,
                  std::vector<std::pair<STDesc, STDesc>> &loop_std_pair, std::vector<LOOP_RESULT> &match_result_list) {
    status = sgtd_shim::SearchLoop(accel_, stds_vec, loop_result, loop_transform, loop_std_pair, match_result_list,
                                   current_frame_id_, config_setting_.candidate_num_, config_setting_.icp_threshold_, CS1);
  }
  void candidate_selector(const std::vector<STDesc> &stds_vec, std::vector<STDMatchList> &candidate_matcher_vec) {
    status = sgtd_shim::candidate_selector(accel_, stds_vec, candidate_matcher_vec, current_frame_id_,
                                           config_setting_.candidate_num_, CS1);
  }
};

int main() {
  ConfigSetting cfg;
  STDescManager mgr(cfg);
  if (mgr.status == SGTD_ERR_NO_DEVICE) {   // CPU-only box: the compile and link check is the test
    std::printf("shim compiled and linked; no gfx950 device here\n");
    return 0;
  }
  if (mgr.status != SGTD_OK) { std::printf("sgtd_create: %s\n", sgtd_strerror(mgr.status)); return 1; }
  // the caller's two loops (semantic_graph_localization.cpp:419-458, :567-604) on a toy map:
  // frames of a 7 x 7 grid of keypoints, shifted a little per frame
  std::vector<std::vector<STDesc>> per_frame;
  for (int f = 0; f < 8; f++) {
    pcl::PointCloud<pcl::PointXYZL>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZL>);
    for (int i = 0; i < 49; i++)
      cloud->points.push_back({(float)(i % 7) * 3.1f + 0.013f * (float)((i * 7 + f) % 11), (float)(i / 7) * 2.7f + 0.017f * (float)((i * 3 + f) % 13),
                               0.1f * (float)((i * 5) % 9), (unsigned)(3 + i % 4)});
    std::vector<STDesc> stds;
    mgr.BuildSingleScanSTD(cloud, stds);
    if (mgr.status != SGTD_OK || stds.empty()) { std::printf("build failed\n"); return 1; }
    if (f < 7) { mgr.AddSTDescs(stds); if (mgr.status != SGTD_OK) return 1; }
    per_frame.push_back(stds);
  }
  if (mgr.current_frame_id_ != 7) { std::printf("frame counter %u\n", mgr.current_frame_id_); return 1; }
  std::pair<int, double> loop_result;
  std::pair<Eigen::Vector3d, Eigen::Matrix3d> loop_transform;
  std::vector<std::pair<STDesc, STDesc>> loop_std_pair;
  std::vector<LOOP_RESULT> match_result_list;
  mgr.SearchLoop(per_frame[7], loop_result, loop_transform, loop_std_pair, match_result_list);
  if (mgr.status != SGTD_OK) { std::printf("SearchLoop: %s\n", sgtd_strerror(mgr.status)); return 1; }
  std::vector<STDMatchList> lists;
  mgr.candidate_selector(per_frame[7], lists);
  if (mgr.status != SGTD_OK || lists.size() != match_result_list.size()) { std::printf("candidate lists differ\n"); return 1; }
  for (size_t i = 0; i < lists.size(); i++)
    if (lists[i].match_id_.first != 7 || lists[i].match_id_.second != match_result_list[i].match_id || lists[i].match_list_.empty()) return 1;
  std::printf("shim ok: %zu candidates, loop (%d, %.0f)\n", lists.size(), loop_result.first, loop_result.second);
  return 0;
}
