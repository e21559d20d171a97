// sgtd_oracle.cpp — CPU ORACLE: a restatement of the reference's triangle
// descriptor build + geometric-hash match path, with the reference's own data
// layout (416-byte AoS descriptor, unordered_map<key, vector<descriptor>>,
// per-j map re-lookups, string based label code) so that timing it is an
// honest stand-in for the reference CPU path.
//
// TEST INFRASTRUCTURE ONLY — see sgtd_oracle.h.  PARITY UNPINNED (no golden
// vectors exist in the reference and it cannot be built here).
//
// Reference locations restated (all under /root/reference/src/sgtd):
//   label code         src/STDesc.cpp:3-16
//   AddSTDescs         src/STDesc.cpp:149-172
//   BuildSingleScanSTD src/STDesc.cpp:174-315
//   candidate_selector src/STDesc.cpp:318-460
//   candidate_verify   src/STDesc.cpp:462-547, triangle_solver :549-571
//   types / constants  include/desc/STDesc.h:31-33,75-97,120-154,217-250
//   input layout       include/utility.hpp:646-659 (xyz f32 + u32 label)
//
// Third-party arithmetic that is NOT in the reference tree and is restated
// from its published behaviour:
//   * PCL KdTreeFLANN::nearestKSearch (FLANN L2_Simple<float>): exact K nearest
//     by f32 squared distance ((dx*dx)+dy*dy)+dz*dz, ascending; ties are
//     implementation defined there, lower index first here.
//   * Eigen 3.3 Vector3d::norm(): sqrt((v0*v0 + v1*v1) + v2*v2) (SSE2 packet
//     over elements 0,1 then the scalar tail), no FMA (reference is built -O3
//     without -march, CMakeLists.txt:5-7).
//   * Eigen::JacobiSVD in triangle_solver: replaced by a one-sided Jacobi SVD
//     written here; only the resulting rotation is compared, to a tolerance.
#include "sgtd_oracle.h"

#include <omp.h>

#include <algorithm>
#include <array>
#include <bitset>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

constexpr int64_t kHashP = 116101;        // HASH_P, STDesc.h:31
constexpr int64_t kMaxN = 10000000000LL;  // MAX_N,  STDesc.h:32

struct V3 {
  double v[3];
  double &operator[](int i) { return v[i]; }
  const double &operator[](int i) const { return v[i]; }
};

inline V3 sub(const V3 &a, const V3 &b) {
  return V3{{a[0] - b[0], a[1] - b[1], a[2] - b[2]}};
}
// Eigen 3.3 fixed-size-3 reduction order, see file header.
inline double norm3(const V3 &a) {
  return std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
}

// Same members, order and size as STDesc (STDesc.h:75-97): 416 bytes.
struct Desc {
  V3 side_length_;
  V3 angle_;
  V3 center_;
  unsigned int frame_id_;
  V3 vertex_A_;
  V3 vertex_B_;
  V3 vertex_C_;
  V3 vertex_attached_;
  std::vector<int> node_id;
  double cov_mat_A_[9];  // [0] carries the global insertion index (oracle only)
  double cov_mat_B_[9];
  double cov_mat_C_[9];
};
static_assert(sizeof(Desc) == 416, "descriptor must keep the reference stride");

// VOXEL_LOC, STDesc.h:126-154
struct MilliKey {
  int64_t x, y, z;
  bool operator==(const MilliKey &o) const {
    return x == o.x && y == o.y && z == o.z;
  }
};
struct MilliKeyHash {
  int64_t operator()(const MilliKey &s) const {
    return (((s.z * kHashP) % kMaxN + s.y) * kHashP) % kMaxN + s.x;
  }
};

// STDesc_LOC, STDesc.h:217-250: six fields, equality on x,y,z,a only.
struct CellKey {
  int64_t x = 0, y = 0, z = 0, a = 0, b = 0, c = 0;
  bool operator==(const CellKey &o) const {
    return x == o.x && y == o.y && z == o.z && a == o.a;
  }
};
struct CellKeyHash {
  int64_t operator()(const CellKey &s) const {
    return ((((((s.z * kHashP) % kMaxN + s.y) * kHashP) % kMaxN + s.x) *
             kHashP) %
                kMaxN +
            s.a);
  }
};

// STDesc.cpp:3-16 — three 4-bit fields through bitset -> string -> stoi.
int label_code(int a, int b, int c) {
  std::bitset<4> ba(a), bb(b), bc(c);
  std::string s = ba.to_string() + bb.to_string() + bc.to_string();
  return std::stoi(s, nullptr, 2);
}

struct P4 {  // pcl::PointXYZL as used: xyz f32 + label u32
  float x, y, z;
  uint32_t label;
};

double ms_between(std::chrono::high_resolution_clock::time_point a,
                  std::chrono::high_resolution_clock::time_point b) {
  return std::chrono::duration_cast<std::chrono::duration<double>>(b - a)
             .count() *
         1000.0;
}

}  // namespace

struct orc_manager {
  orc_config cfg;
  unsigned int current_frame_id_ = 0;
  std::unordered_map<CellKey, std::vector<Desc>, CellKeyHash> data_base_;
  int64_t n_entries = 0;
  // global insertion index -> (key, position in bucket)
  std::vector<std::pair<CellKey, int64_t>> entry_loc;

  std::vector<Desc> last;  // last BuildSingleScanSTD result

  // results of the last candidate_selector
  std::vector<Desc> last_query;
  std::vector<int32_t> cand_frame, cand_votes;
  std::vector<int64_t> cand_off;
  std::vector<int32_t> cm_q;
  std::vector<int64_t> cm_e;
  std::vector<int32_t> rm_q, rm_cell, rm_j;
  std::vector<int64_t> rm_e;
  std::vector<uint32_t> rm_frame;
  std::vector<double> rm_dis;
  std::vector<double> votes;
  orc_counters cnt{};
};

// ---------------------------------------------------------------------------
// BuildSingleScanSTD — STDesc.cpp:174-315
// ---------------------------------------------------------------------------
static void build_single_scan(orc_manager *mg, const std::vector<P4> &pc,
                              std::vector<Desc> &out) {
  out.clear();
  const double scale = 1.0 / mg->cfg.std_side_resolution;  // :178
  const int near_num = mg->cfg.descriptor_near_num;
  const double max_len = mg->cfg.descriptor_max_len;
  const double min_len = mg->cfg.descriptor_min_len;
  std::unordered_map<MilliKey, bool, MilliKeyHash> seen;  // feat_map :182
  const int n = (int)pc.size();
  // Contract: the reference indexes K results even when
  // the cloud holds fewer than K points (UB); such frames yield no descriptor.
  if (n < near_num) return;

  std::vector<int> nn_idx(near_num);
  std::vector<float> nn_d(near_num);
  std::vector<std::pair<float, int>> all(n);
  for (int i = 0; i < n; i++) {
    const P4 sp = pc[i];
    // exact K nearest, f32 L2_Simple accumulation order, ties -> lower index
    for (int j = 0; j < n; j++) {
      float dx = sp.x - pc[j].x, dy = sp.y - pc[j].y, dz = sp.z - pc[j].z;
      float d = dx * dx;
      d += dy * dy;
      d += dz * dz;
      all[j] = {d, j};
    }
    std::partial_sort(all.begin(), all.begin() + near_num, all.end());
    for (int k = 0; k < near_num; k++) {
      nn_idx[k] = all[k].second;
      nn_d[k] = all[k].first;
    }
    for (int m = 1; m < near_num - 1; m++) {        // :193
      for (int nn = m + 1; nn < near_num; nn++) {   // :194
        const P4 p1 = sp, p2 = pc[nn_idx[m]], p3 = pc[nn_idx[nn]];
        // float subtraction first, then pow(double,2) (== exact square), two
        // rounded adds and a correctly rounded sqrt — :198-203
        double a = std::sqrt(std::pow(p1.x - p2.x, 2) + std::pow(p1.y - p2.y, 2) +
                             std::pow(p1.z - p2.z, 2));
        double b = std::sqrt(std::pow(p1.x - p3.x, 2) + std::pow(p1.y - p3.y, 2) +
                             std::pow(p1.z - p3.z, 2));
        double c = std::sqrt(std::pow(p3.x - p2.x, 2) + std::pow(p3.y - p2.y, 2) +
                             std::pow(p3.z - p2.z, 2));
        if (a > max_len || b > max_len || c > max_len || a < min_len ||
            b < min_len || c < min_len)
          continue;  // :204-208
        // incidence triples: l1 = side a touches (p1,p2), l2 = b (p1,p3),
        // l3 = c (p2,p3) — :213-218
        int l1[3] = {1, 2, 0}, l2[3] = {1, 0, 3}, l3[3] = {0, 2, 3};
        auto swap_side = [](double &u, double &v, int *lu, int *lv) {
          std::swap(u, v);
          for (int k = 0; k < 3; k++) std::swap(lu[k], lv[k]);
        };
        if (a > b) swap_side(a, b, l1, l2);  // :220-227
        if (b > c) swap_side(b, c, l2, l3);  // :228-235
        if (a > b) swap_side(a, b, l1, l2);  // :236-243
        // millimetre key through a float (pcl::PointXYZ members) — :246-250
        float kx = a * 1000, ky = b * 1000, kz = c * 1000;
        MilliKey mk{(int64_t)kx, (int64_t)ky, (int64_t)kz};
        if (seen.find(mk) != seen.end()) continue;  // :251-253
        V3 A, B, C, lab;
        auto pick = [&](const int *u, const int *v, V3 &dst, double &ldst) {
          const P4 *p = (u[0] == v[0]) ? &p1 : (u[1] == v[1]) ? &p2 : &p3;
          dst = V3{{(double)p->x, (double)p->y, (double)p->z}};
          ldst = (double)p->label;
        };
        pick(l1, l2, A, lab[0]);  // :255-267
        pick(l1, l3, B, lab[1]);  // :268-280
        pick(l2, l3, C, lab[2]);  // :281-293
        Desc d{};
        d.vertex_A_ = A;
        d.vertex_B_ = B;
        d.vertex_C_ = C;
        for (int k = 0; k < 3; k++) d.center_[k] = ((A[k] + B[k]) + C[k]) / 3;
        d.vertex_attached_ = lab;
        d.side_length_ = V3{{scale * a, scale * b, scale * c}};
        d.angle_[0] = std::fabs((b * b + c * c - a * a) / (2 * b * c));
        d.angle_[1] = std::fabs((a * a + c * c - b * b) / (2 * a * c));
        d.angle_[2] = std::fabs((a * a + b * b - c * c) / (2 * a * b));
        d.node_id = std::vector<int>{i, m, nn};  // m,n are ranks (:302-303)
        d.frame_id_ = mg->current_frame_id_;     // :305
        seen[mk] = true;
        out.push_back(d);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// AddSTDescs — STDesc.cpp:149-172
// ---------------------------------------------------------------------------
static void add_descs(orc_manager *mg, const std::vector<Desc> &v) {
  mg->current_frame_id_++;  // :151, before the loop
  for (auto single : v) {   // by-value copy, as the reference
    CellKey pos;
    pos.x = (int)(single.side_length_[0] + 0.5);
    pos.y = (int)(single.side_length_[1] + 0.5);
    pos.z = (int)(single.side_length_[2] + 0.5);
    pos.a = (int)(single.vertex_attached_[0]);
    pos.b = (int)(single.vertex_attached_[1]);
    pos.c = (int)(single.vertex_attached_[2]);
    pos.a = label_code((int)pos.a, (int)pos.b, (int)pos.c);
    single.cov_mat_A_[0] = (double)mg->n_entries;  // oracle bookkeeping
    auto it = mg->data_base_.find(pos);
    int64_t j;
    if (it != mg->data_base_.end()) {
      j = (int64_t)mg->data_base_[pos].size();
      mg->data_base_[pos].push_back(single);
    } else {
      std::vector<Desc> bucket;
      bucket.push_back(single);
      mg->data_base_[pos] = bucket;
      j = 0;
    }
    mg->entry_loc.push_back({pos, j});
    mg->n_entries++;
  }
}

// ---------------------------------------------------------------------------
// candidate_selector — STDesc.cpp:318-460
// ---------------------------------------------------------------------------
static void select_candidates(orc_manager *mg, const std::vector<Desc> &q) {
  auto t1 = std::chrono::high_resolution_clock::now();
  const int max_frame_n = mg->cfg.max_frame_n;
  std::vector<double> match_array(max_frame_n, 0.0);  // :323 (stack array there)
  std::vector<int> match_index_vec;
  int voxel_round[27][3];  // :326-334, x outer .. z inner
  {
    int k = 0;
    for (int x = -1; x <= 1; x++)
      for (int y = -1; y <= 1; y++)
        for (int z = -1; z <= 1; z++) {
          voxel_round[k][0] = x;
          voxel_round[k][1] = y;
          voxel_round[k][2] = z;
          k++;
        }
  }
  const size_t nq = q.size();
  std::vector<char> useful(nq, 0);
  std::vector<std::vector<size_t>> useful_index(nq);
  std::vector<std::vector<CellKey>> useful_pos(nq);
  std::vector<std::vector<int>> useful_cell(nq);    // oracle bookkeeping
  std::vector<std::vector<double>> useful_dis(nq);  // oracle bookkeeping
  int64_t visited = 0;
  const double rough = mg->cfg.rough_dis_threshold;
  auto &db = mg->data_base_;

  omp_set_num_threads(mg->cfg.num_threads > 0 ? mg->cfg.num_threads : 1);  // :348
#pragma omp parallel for reduction(+ : visited)
  for (size_t i = 0; i < nq; i++) {
    Desc src = q[i];  // deep copy, :352
    CellKey pos;
    double dis_threshold = norm3(src.side_length_) * rough;  // :356-357
    for (int vi = 0; vi < 27; vi++) {
      pos.x = (int)(src.side_length_[0] + voxel_round[vi][0]);  // trunc, :359
      pos.y = (int)(src.side_length_[1] + voxel_round[vi][1]);
      pos.z = (int)(src.side_length_[2] + voxel_round[vi][2]);
      pos.a = (int)(src.vertex_attached_[0]);
      pos.b = (int)(src.vertex_attached_[1]);
      pos.c = (int)(src.vertex_attached_[2]);
      pos.a = label_code((int)pos.a, (int)pos.b, (int)pos.c);  // :365
      V3 centre{{(double)pos.x + 0.5, (double)pos.y + 0.5, (double)pos.z + 0.5}};
      if (norm3(sub(src.side_length_, centre)) < 1.5) {  // :369
        auto it = db.find(pos);
        if (it != db.end()) {
          for (size_t j = 0; j < db[pos].size(); j++) {  // :372, re-hash per j
            visited++;
            // unsigned subtraction: true iff the ids differ — :373
            if ((src.frame_id_ - db[pos][j].frame_id_) > 0) {
              double dis = norm3(sub(src.side_length_, db[pos][j].side_length_));
              if (dis < dis_threshold) {  // :378
                useful[i] = 1;
                useful_pos[i].push_back(pos);
                useful_index[i].push_back(j);
                useful_cell[i].push_back(vi);
                useful_dis[i].push_back(dis);
              }
            }
          }
        }
      }
    }
  }
  auto t2 = std::chrono::high_resolution_clock::now();

  // vote accumulation — :404-420
  std::vector<std::pair<int, int>> index_recorder;
  mg->rm_q.clear(); mg->rm_cell.clear(); mg->rm_j.clear();
  mg->rm_e.clear(); mg->rm_frame.clear(); mg->rm_dis.clear();
  for (size_t i = 0; i < nq; i++) {
    if (!useful[i]) continue;
    for (size_t j = 0; j < useful_index[i].size(); j++) {
      const Desc &e = db[useful_pos[i][j]][useful_index[i][j]];
      if ((int64_t)e.frame_id_ < (int64_t)max_frame_n)  // ref: OOB beyond
        match_array[e.frame_id_] += 1;
      index_recorder.push_back({(int)i, (int)j});
      match_index_vec.push_back((int)e.frame_id_);
      mg->rm_q.push_back((int32_t)i);
      mg->rm_cell.push_back(useful_cell[i][j]);
      mg->rm_j.push_back((int32_t)useful_index[i][j]);
      mg->rm_e.push_back((int64_t)e.cov_mat_A_[0]);
      mg->rm_frame.push_back(e.frame_id_);
      mg->rm_dis.push_back(useful_dis[i][j]);
    }
  }
  mg->votes = match_array;

  // top candidate_num frames — :423-453
  mg->cand_frame.clear(); mg->cand_votes.clear();
  mg->cand_off.assign(1, 0);
  mg->cm_q.clear(); mg->cm_e.clear();
  for (int cnt = 0; cnt < mg->cfg.candidate_num; cnt++) {
    double max_vote = 1;
    int max_vote_index = -1;
    for (int i = 0; i < max_frame_n; i++) {
      if (match_array[i] > max_vote) {
        max_vote = match_array[i];
        max_vote_index = i;
      }
    }
    if (max_vote_index >= 0 && max_vote >= 5) {
      match_array[max_vote_index] = 0;
      for (size_t i = 0; i < index_recorder.size(); i++) {
        if (match_index_vec[i] == max_vote_index) {
          // the reference copies both 416-byte descriptors here (:441-448)
          std::pair<Desc, Desc> pr;
          pr.first = q[index_recorder[i].first];
          pr.second = db[useful_pos[index_recorder[i].first][index_recorder[i].second]]
                        [useful_index[index_recorder[i].first][index_recorder[i].second]];
          mg->cm_q.push_back(index_recorder[i].first);
          mg->cm_e.push_back((int64_t)pr.second.cov_mat_A_[0]);
        }
      }
      mg->cand_frame.push_back(max_vote_index);
      mg->cand_votes.push_back((int32_t)max_vote);
      mg->cand_off.push_back((int64_t)mg->cm_q.size());
    } else {
      break;
    }
  }
  auto t4 = std::chrono::high_resolution_clock::now();
  mg->cnt.D = (int64_t)nq;
  mg->cnt.P = visited;
  mg->cnt.M = (int64_t)mg->rm_q.size();
  mg->cnt.probe_ms = ms_between(t1, t2);
  mg->cnt.select_ms = ms_between(t1, t4);
}

// ---------------------------------------------------------------------------
// candidate_verify / triangle_solver — STDesc.cpp:462-571 ("next" row)
// ---------------------------------------------------------------------------
namespace {

struct M3 {
  double m[3][3];
};

M3 mul(const M3 &a, const M3 &b) {
  M3 r{};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
M3 transpose(const M3 &a) {
  M3 r{};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
double det(const M3 &a) {
  return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) -
         a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
         a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}

// One-sided (Hestenes) Jacobi SVD of a 3x3: H = U diag(s) V^T.  Columns that
// belong to (near) zero singular values are completed by cross products so
// that U and V are orthogonal matrices.
void svd3(const M3 &H, M3 &U, M3 &V) {
  M3 A = H;
  M3 W{};
  for (int i = 0; i < 3; i++) W.m[i][i] = 1;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 3; k++) {
          alpha += A.m[k][p] * A.m[k][p];
          beta += A.m[k][q] * A.m[k][q];
          gamma += A.m[k][p] * A.m[k][q];
        }
        off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
        if (std::fabs(gamma) < 1e-300) continue;
        double zeta = (beta - alpha) / (2 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        double c = 1 / std::sqrt(1 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) {
          double ap = A.m[k][p], aq = A.m[k][q];
          A.m[k][p] = c * ap - s * aq;
          A.m[k][q] = s * ap + c * aq;
          double wp = W.m[k][p], wq = W.m[k][q];
          W.m[k][p] = c * wp - s * wq;
          W.m[k][q] = s * wp + c * wq;
        }
      }
    if (off < 1e-15) break;
  }
  double s[3];
  int order[3] = {0, 1, 2};
  for (int j = 0; j < 3; j++)
    s[j] = std::sqrt(A.m[0][j] * A.m[0][j] + A.m[1][j] * A.m[1][j] + A.m[2][j] * A.m[2][j]);
  std::sort(order, order + 3, [&](int a, int b) { return s[a] > s[b]; });
  double smax = s[order[0]];
  int rank = 0;
  for (int jj = 0; jj < 3; jj++) {
    int j = order[jj];
    for (int k = 0; k < 3; k++) V.m[k][jj] = W.m[k][j];
    if (s[j] > 1e-12 * (smax > 0 ? smax : 1)) {
      for (int k = 0; k < 3; k++) U.m[k][jj] = A.m[k][j] / s[j];
      rank = jj + 1;
    }
  }
  auto cross_into = [](M3 &X, int a, int b, int dst) {
    X.m[0][dst] = X.m[1][a] * X.m[2][b] - X.m[2][a] * X.m[1][b];
    X.m[1][dst] = X.m[2][a] * X.m[0][b] - X.m[0][a] * X.m[2][b];
    X.m[2][dst] = X.m[0][a] * X.m[1][b] - X.m[1][a] * X.m[0][b];
  };
  if (rank == 2) {
    cross_into(U, 0, 1, 2);
  } else if (rank < 2) {
    // degenerate (collinear / coincident triangle): any completion; keep
    // identity-like columns so the result is at least orthogonal
    for (int i = 0; i < 3; i++)
      for (int j = rank; j < 3; j++) U.m[i][j] = (i == j) ? 1 : 0;
  }
}

void solve_triangle(const Desc &qa, const Desc &db, double t[3], M3 &rot) {
  M3 src{}, ref{};
  const V3 *qs[3] = {&qa.vertex_A_, &qa.vertex_B_, &qa.vertex_C_};
  const V3 *ds[3] = {&db.vertex_A_, &db.vertex_B_, &db.vertex_C_};
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) {
      src.m[r][c] = (*qs[c])[r] - qa.center_[r];
      ref.m[r][c] = (*ds[c])[r] - db.center_[r];
    }
  M3 cov = mul(src, transpose(ref));  // :558
  M3 U{}, V{};
  svd3(cov, U, V);
  rot = mul(V, transpose(U));  // :563
  if (det(rot) < 0) {          // :564-568
    M3 K{};
    K.m[0][0] = 1; K.m[1][1] = 1; K.m[2][2] = -1;
    rot = mul(mul(V, K), transpose(U));
  }
  for (int r = 0; r < 3; r++)  // :569
    t[r] = -(rot.m[r][0] * qa.center_[0] + rot.m[r][1] * qa.center_[1] +
             rot.m[r][2] * qa.center_[2]) +
           db.center_[r];
}

inline bool pair_inlier(const Desc &qa, const Desc &db, const double t[3],
                        const M3 &rot, double thr) {
  const V3 *qs[3] = {&qa.vertex_A_, &qa.vertex_B_, &qa.vertex_C_};
  const V3 *ds[3] = {&db.vertex_A_, &db.vertex_B_, &db.vertex_C_};
  for (int v = 0; v < 3; v++) {
    V3 p;
    for (int r = 0; r < 3; r++)
      p[r] = (rot.m[r][0] * (*qs[v])[0] + rot.m[r][1] * (*qs[v])[1] +
              rot.m[r][2] * (*qs[v])[2]) + t[r];
    if (!(norm3(sub(p, *ds[v])) < thr)) return false;
  }
  return true;
}

}  // namespace

// ---------------------------------------------------------------------------
// C interface
// ---------------------------------------------------------------------------
static void desc_from_soa(const orc_desc_soa *d, int64_t i, Desc &o) {
  for (int k = 0; k < 3; k++) {
    o.side_length_[k] = d->side[i * 3 + k];
    o.angle_[k] = d->angle ? d->angle[i * 3 + k] : 0;
    o.center_[k] = d->center ? d->center[i * 3 + k] : 0;
    o.vertex_A_[k] = d->vertex ? d->vertex[i * 9 + k] : 0;
    o.vertex_B_[k] = d->vertex ? d->vertex[i * 9 + 3 + k] : 0;
    o.vertex_C_[k] = d->vertex ? d->vertex[i * 9 + 6 + k] : 0;
    o.vertex_attached_[k] = (double)d->label[i * 3 + k];
  }
  o.frame_id_ = d->frame[i];
  if (d->node_id)
    o.node_id = {d->node_id[i * 3], d->node_id[i * 3 + 1], d->node_id[i * 3 + 2]};
  else
    o.node_id = {0, 0, 0};
}
static void desc_to_soa(const Desc &s, orc_desc_soa *d, int64_t i) {
  for (int k = 0; k < 3; k++) {
    if (d->side) d->side[i * 3 + k] = s.side_length_[k];
    if (d->angle) d->angle[i * 3 + k] = s.angle_[k];
    if (d->center) d->center[i * 3 + k] = s.center_[k];
    if (d->vertex) {
      d->vertex[i * 9 + k] = s.vertex_A_[k];
      d->vertex[i * 9 + 3 + k] = s.vertex_B_[k];
      d->vertex[i * 9 + 6 + k] = s.vertex_C_[k];
    }
    if (d->label) d->label[i * 3 + k] = (int32_t)s.vertex_attached_[k];
    if (d->node_id) d->node_id[i * 3 + k] = s.node_id.size() == 3 ? s.node_id[k] : 0;
  }
  if (d->frame) d->frame[i] = s.frame_id_;
}

extern "C" {

orc_manager *orc_create(const orc_config *cfg) {
  orc_manager *m = new orc_manager();
  m->cfg = *cfg;
  m->current_frame_id_ = 0;  // STDesc.h:363
  return m;
}
void orc_destroy(orc_manager *m) { delete m; }
uint32_t orc_current_frame_id(const orc_manager *m) { return m->current_frame_id_; }
void orc_set_current_frame_id(orc_manager *m, uint32_t id) { m->current_frame_id_ = id; }
// timing protocol of BASELINE.md §2: the same table timed at several OpenMP thread settings
void orc_set_num_threads(orc_manager *m, int n) { m->cfg.num_threads = n > 0 ? n : 1; }
int orc_label_code(int a, int b, int c) { return label_code(a, b, c); }
// the restated key types, for the pin against the reference's own (oracle/ref_pin.cpp)
int orc_cell_key_eq(const int64_t *p, const int64_t *q) {
  CellKey a, b;
  a.x = p[0]; a.y = p[1]; a.z = p[2]; a.a = p[3]; a.b = p[4]; a.c = p[5];
  b.x = q[0]; b.y = q[1]; b.z = q[2]; b.a = q[3]; b.b = q[4]; b.c = q[5];
  return a == b;
}
int64_t orc_cell_key_hash(const int64_t *p) {
  CellKey a;
  a.x = p[0]; a.y = p[1]; a.z = p[2]; a.a = p[3]; a.b = p[4]; a.c = p[5];
  return CellKeyHash()(a);
}
int orc_milli_key_eq(const int64_t *p, const int64_t *q) { return MilliKey{p[0], p[1], p[2]} == MilliKey{q[0], q[1], q[2]}; }
int64_t orc_milli_key_hash(const int64_t *p) { return MilliKeyHash()(MilliKey{p[0], p[1], p[2]}); }

int64_t orc_build(orc_manager *m, const float *xyz, const uint32_t *label, int n) {
  std::vector<P4> pc(n);
  for (int i = 0; i < n; i++)  // Graph2CloudL, utility.hpp:646-659
    pc[i] = P4{xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], label[i]};
  auto t0 = std::chrono::high_resolution_clock::now();
  build_single_scan(m, pc, m->last);
  auto t1 = std::chrono::high_resolution_clock::now();
  m->cnt.build_ms = ms_between(t0, t1);
  return (int64_t)m->last.size();
}
void orc_last_export(const orc_manager *m, orc_desc_soa *out) {
  for (size_t i = 0; i < m->last.size(); i++) desc_to_soa(m->last[i], out, (int64_t)i);
}
void orc_add_last(orc_manager *m) { add_descs(m, m->last); }
// n_frames frames of n keypoints each, built and inserted one after the other exactly like n_frames
// (orc_build, orc_add_last) pairs: the builds — independent of each other but for the frame id they
// stamp, current_frame_id_ + f — run on all host threads, the inserts stay serial and in order.
// (test infrastructure: a 10 000-frame oracle map in seconds instead of half a minute)
void orc_add_frames(orc_manager *m, const float *xyz, const uint32_t *label, int n_frames, int n) {
  std::vector<std::vector<Desc>> built((size_t)n_frames);
  const unsigned int id0 = m->current_frame_id_;
#pragma omp parallel for schedule(dynamic, 8)
  for (int f = 0; f < n_frames; f++) {
    std::vector<P4> pc(n);
    for (int i = 0; i < n; i++)
      pc[i] = P4{xyz[((size_t)f * n + i) * 3], xyz[((size_t)f * n + i) * 3 + 1], xyz[((size_t)f * n + i) * 3 + 2], label[(size_t)f * n + i]};
    orc_manager local = orc_manager();      // only cfg and the frame id are read by the build
    local.cfg = m->cfg;
    local.current_frame_id_ = id0 + (unsigned int)f;
    build_single_scan(&local, pc, built[(size_t)f]);
  }
  for (int f = 0; f < n_frames; f++) add_descs(m, built[(size_t)f]);
  if (n_frames > 0) m->last = built[(size_t)n_frames - 1];
}
void orc_add(orc_manager *m, const orc_desc_soa *d, int64_t n) {
  std::vector<Desc> v(n);
  for (int64_t i = 0; i < n; i++) desc_from_soa(d, i, v[i]);
  add_descs(m, v);
}

int orc_select(orc_manager *m, int use_last, const orc_desc_soa *q, int64_t nq,
               int32_t *cand_frame, int32_t *cand_votes, int64_t *cand_off,
               int32_t *n_cand) {
  if (use_last) {
    m->last_query = m->last;
  } else {
    m->last_query.assign(nq, Desc{});
    for (int64_t i = 0; i < nq; i++) desc_from_soa(q, i, m->last_query[i]);
  }
  select_candidates(m, m->last_query);
  int nc = (int)m->cand_frame.size();
  if (n_cand) *n_cand = nc;
  for (int i = 0; i < nc; i++) {
    if (cand_frame) cand_frame[i] = m->cand_frame[i];
    if (cand_votes) cand_votes[i] = m->cand_votes[i];
  }
  if (cand_off)
    for (int i = 0; i <= nc; i++) cand_off[i] = m->cand_off[i];
  return 0;
}
int64_t orc_cand_match_total(const orc_manager *m) { return (int64_t)m->cm_q.size(); }
void orc_cand_matches(const orc_manager *m, int32_t *q_idx, int64_t *db_entry) {
  for (size_t i = 0; i < m->cm_q.size(); i++) {
    q_idx[i] = m->cm_q[i];
    db_entry[i] = m->cm_e[i];
  }
}
int64_t orc_rough_total(const orc_manager *m) { return (int64_t)m->rm_q.size(); }
void orc_rough_matches(const orc_manager *m, int32_t *q_idx, int32_t *cell,
                       int32_t *j, int64_t *db_entry, uint32_t *frame, double *dis) {
  for (size_t i = 0; i < m->rm_q.size(); i++) {
    if (q_idx) q_idx[i] = m->rm_q[i];
    if (cell) cell[i] = m->rm_cell[i];
    if (j) j[i] = m->rm_j[i];
    if (db_entry) db_entry[i] = m->rm_e[i];
    if (frame) frame[i] = m->rm_frame[i];
    if (dis) dis[i] = m->rm_dis[i];
  }
}
void orc_votes(const orc_manager *m, double *votes) {
  std::memcpy(votes, m->votes.data(), m->votes.size() * sizeof(double));
}

void orc_fetch_entries(const orc_manager *m, const int64_t *db_entry, int64_t n,
                       orc_desc_soa *out) {
  for (int64_t i = 0; i < n; i++) {
    const auto &loc = m->entry_loc[db_entry[i]];
    const Desc &d = m->data_base_.at(loc.first)[loc.second];
    desc_to_soa(d, out, i);
  }
}

void orc_table_dump(const orc_manager *m, int64_t *keys, int64_t *bucket_off,
                    int64_t *entry_ids) {
  std::vector<const std::pair<const CellKey, std::vector<Desc>> *> rows;
  for (auto &kv : m->data_base_) rows.push_back(&kv);
  std::sort(rows.begin(), rows.end(), [](auto *l, auto *r) {
    const CellKey &a = l->first, &b = r->first;
    if (a.a != b.a) return a.a < b.a;
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.z < b.z;
  });
  int64_t off = 0;
  for (size_t u = 0; u < rows.size(); u++) {
    keys[u * 4 + 0] = rows[u]->first.x;
    keys[u * 4 + 1] = rows[u]->first.y;
    keys[u * 4 + 2] = rows[u]->first.z;
    keys[u * 4 + 3] = rows[u]->first.a;
    bucket_off[u] = off;
    for (const Desc &d : rows[u]->second) entry_ids[off++] = (int64_t)d.cov_mat_A_[0];
  }
  bucket_off[rows.size()] = off;
}

void orc_get_counters(const orc_manager *m, orc_counters *c) {
  *c = m->cnt;
  c->E = m->n_entries;
  c->U = (int64_t)m->data_base_.size();
}

double orc_verify(orc_manager *m, int cand, double *t_out, double *rot_out,
                  int32_t *success_idx, int32_t *n_success) {
  // rebuild match_list_ of this candidate as (query desc, table desc) pairs
  const int64_t lo = m->cand_off[cand], hi = m->cand_off[cand + 1];
  const int64_t n = hi - lo;
  std::vector<std::pair<const Desc *, const Desc *>> ml(n);
  for (int64_t k = 0; k < n; k++) {
    const auto &loc = m->entry_loc[m->cm_e[lo + k]];
    ml[k] = {&m->last_query[m->cm_q[lo + k]], &m->data_base_.at(loc.first)[loc.second]};
  }
  const int skip_len = (int)(n / 50) + 1;  // :467
  const int use_size = (int)(n / skip_len);  // :468
  const double dis_threshold = 3.0;          // :469
  std::vector<int> vote_list(use_size);
  omp_set_num_threads(m->cfg.num_threads > 0 ? m->cfg.num_threads : 1);
#pragma omp parallel for
  for (int i = 0; i < use_size; i++) {  // :481-506
    double tt[3];
    M3 rr;
    solve_triangle(*ml[(size_t)i * skip_len].first, *ml[(size_t)i * skip_len].second, tt, rr);
    int vote = 0;
    for (int64_t j = 0; j < n; j++)
      if (pair_inlier(*ml[j].first, *ml[j].second, tt, rr, dis_threshold)) vote++;
    vote_list[i] = vote;
  }
  int max_vote_index = 0, max_vote = 0;  // :507-514, first maximum wins
  for (int i = 0; i < use_size; i++)
    if (max_vote < vote_list[i]) {
      max_vote_index = i;
      max_vote = vote_list[i];
    }
  if (n_success) *n_success = 0;
  if (max_vote >= 4) {  // :515
    double tt[3];
    M3 rr;
    solve_triangle(*ml[(size_t)max_vote_index * skip_len].first,
                   *ml[(size_t)max_vote_index * skip_len].second, tt, rr);
    int ns = 0;
    for (int64_t j = 0; j < n; j++)
      if (pair_inlier(*ml[j].first, *ml[j].second, tt, rr, dis_threshold)) {
        if (success_idx) success_idx[ns] = (int32_t)j;
        ns++;
      }
    if (n_success) *n_success = ns;
    if (t_out)
      for (int r = 0; r < 3; r++) t_out[r] = tt[r];
    if (rot_out)
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) rot_out[r * 3 + c] = rr.m[r][c];
    return (double)ns;  // :539
  }
  return -1;  // :541
}

// ---------------------------------------------------------------------------
// parity-risk audit (sgtd_oracle.h): the same loops with every inferred piece of arithmetic evaluated in its
// plausible alternatives side by side
// ---------------------------------------------------------------------------
namespace audit {
// sum of three squares, reference form and the three variants of sgtd_oracle.h
inline double ss_ref(double a, double b, double c) { return (a * a + b * b) + c * c; }
inline double ss_var(int k, double a, double b, double c) {
  if (k == 0) return a * a + (b * b + c * c);
  if (k == 1) return std::fma(c, c, std::fma(b, b, a * a));
  return std::fma(a, a, std::fma(b, b, c * c));
}
inline double ulp_of(double x) { return std::nextafter(std::fabs(x), INFINITY) - std::fabs(x); }
inline void lower(double &m, double v) { if (v < m) m = v; }
}  // namespace audit

void orc_audit_build(orc_manager *mg, const float *xyz, const uint32_t *label, int n, orc_audit *A) {
  using namespace audit;
  orc_build(mg, xyz, label, n);
  const int K = mg->cfg.descriptor_near_num;
  if (n < K) return;
  const double scale = 1.0 / mg->cfg.std_side_resolution, max_len = mg->cfg.descriptor_max_len, min_len = mg->cfg.descriptor_min_len;
  std::vector<P4> pc(n);
  for (int i = 0; i < n; i++) pc[i] = P4{xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], label[i]};
  std::vector<std::pair<float, int>> all(n), allf(n);
  std::vector<int> nn(K);
  const int kk = std::min(K + 1, n);
  for (int i = 0; i < n; i++) {
    const P4 sp = pc[i];
    for (int j = 0; j < n; j++) {
      const float dx = sp.x - pc[j].x, dy = sp.y - pc[j].y, dz = sp.z - pc[j].z;
      float d = dx * dx;
      d += dy * dy;
      d += dz * dz;
      all[j] = {d, j};
      allf[j] = {std::fma(dz, dz, std::fma(dy, dy, dx * dx)), j};
    }
    std::partial_sort(all.begin(), all.begin() + kk, all.end());
    std::partial_sort(allf.begin(), allf.begin() + kk, allf.end());
    A->knn_points++;
    bool tie = false, order = false;
    for (int k = 0; k + 1 < kk; k++) tie = tie || all[k].first == all[k + 1].first;
    for (int k = 0; k < K; k++) order = order || all[k].second != allf[k].second;
    A->knn_tied_points += tie;
    A->knn_fma_order_diffs += order;
    for (int k = 0; k < K; k++) nn[k] = all[k].second;
    for (int m = 1; m < K - 1; m++)
      for (int q = m + 1; q < K; q++) {
        const P4 p1 = sp, p2 = pc[nn[m]], p3 = pc[nn[q]];
        // the three sides: f32 differences (exact as doubles), then the sum of squares in the reference form and its variants
        const double d[3][3] = {{(double)(p1.x - p2.x), (double)(p1.y - p2.y), (double)(p1.z - p2.z)},
                                {(double)(p1.x - p3.x), (double)(p1.y - p3.y), (double)(p1.z - p3.z)},
                                {(double)(p3.x - p2.x), (double)(p3.y - p2.y), (double)(p3.z - p2.z)}};
        double ref[3];
        for (int s = 0; s < 3; s++) ref[s] = std::sqrt(ss_ref(d[s][0], d[s][1], d[s][2]));
        A->triplets++;
        // the integer decisions a triplet takes part in, as one signature
        auto signature = [&](const double *v, bool margins) {
          unsigned long long sig = 0;
          for (int s = 0; s < 3; s++) {
            sig = sig * 4 + (v[s] > max_len ? 1 : 0) * 2 + (v[s] < min_len ? 1 : 0);
            if (margins) { lower(A->min_len_margin, std::fabs(v[s] - max_len)); lower(A->min_len_margin, std::fabs(v[s] - min_len)); }
          }
          double a = v[0], b = v[1], c = v[2];
          sig = sig * 2 + (a > b);
          if (a > b) std::swap(a, b);
          sig = sig * 2 + (b > c);
          if (b > c) std::swap(b, c);
          sig = sig * 2 + (a > b);
          if (a > b) std::swap(a, b);
          const double srt[3] = {a, b, c};
          unsigned long long h = 1469598103934665603ull;
          for (int s = 0; s < 3; s++) {
            const float kf = (float)(srt[s] * 1000);
            const double sc = scale * srt[s];
            const long long parts[3] = {(long long)kf, (long long)(int)(sc + 0.5), (long long)(int)sc};
            for (long long pv : parts) h = (h ^ (unsigned long long)pv) * 1099511628211ull;
            if (margins) {
              const double fr = sc - std::floor(sc);
              lower(A->min_cell_margin, std::min(fr, 1.0 - fr));
              lower(A->min_cell_margin, std::fabs(fr - 0.5));
            }
          }
          return sig * 1000003ull + h;
        };
        const unsigned long long sig_ref = signature(ref, true);
        for (int k = 0; k < 3; k++) {
          double var[3];
          for (int s = 0; s < 3; s++) {
            var[s] = std::sqrt(ss_var(k, d[s][0], d[s][1], d[s][2]));
            A->side_value_diffs[k] += var[s] != ref[s];
          }
          A->build_flips[k] += signature(var, false) != sig_ref;
        }
      }
  }
}

void orc_audit_select(orc_manager *mg, orc_audit *A) {
  using namespace audit;
  const std::vector<Desc> &q = mg->last;
  const double rough = mg->cfg.rough_dis_threshold;
  auto &db = mg->data_base_;
  omp_set_num_threads(mg->cfg.num_threads > 0 ? mg->cfg.num_threads : 1);
  const size_t nq = q.size();
  std::vector<orc_audit> part((size_t)omp_get_max_threads());
  for (auto &p : part) { std::memset(&p, 0, sizeof p); p.min_margin = p.min_margin_ulps = p.min_gate_margin = INFINITY; }
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t i = 0; i < nq; i++) {
    orc_audit &T = part[(size_t)omp_get_thread_num()];
    const Desc &src = q[i];
    const double s0 = src.side_length_[0], s1 = src.side_length_[1], s2 = src.side_length_[2];
    const double thr = std::sqrt(ss_ref(s0, s1, s2)) * rough;
    double thr_v[3];
    for (int k = 0; k < 3; k++) thr_v[k] = std::sqrt(ss_var(k, s0, s1, s2)) * rough;
    const double thr_up = std::nextafter(thr, INFINITY), thr_dn = std::nextafter(thr, -INFINITY);
    const double u = ulp_of(thr);
    CellKey pos;
    pos.a = label_code((int)src.vertex_attached_[0], (int)src.vertex_attached_[1], (int)src.vertex_attached_[2]);
    for (int x = -1; x <= 1; x++)
      for (int y = -1; y <= 1; y++)
        for (int z = -1; z <= 1; z++) {
          pos.x = (int)(s0 + x); pos.y = (int)(s1 + y); pos.z = (int)(s2 + z);
          const double c0 = s0 - ((double)pos.x + 0.5), c1 = s1 - ((double)pos.y + 0.5), c2 = s2 - ((double)pos.z + 0.5);
          const double gd = std::sqrt(ss_ref(c0, c1, c2));
          const bool gate = gd < 1.5;
          T.gate_tests++;
          lower(T.min_gate_margin, std::fabs(gd - 1.5));
          auto it = db.find(pos);
          const int64_t len = it == db.end() ? 0 : (int64_t)it->second.size();
          for (int k = 0; k < 3; k++)
            if ((std::sqrt(ss_var(k, c0, c1, c2)) < 1.5) != gate) { T.gate_flips[k]++; T.gate_flip_visits[k] += len; }
          if (!gate || it == db.end()) continue;
          for (const Desc &e : it->second) {
            if (!((src.frame_id_ - e.frame_id_) > 0)) continue;
            const double d0 = s0 - e.side_length_[0], d1 = s1 - e.side_length_[1], d2 = s2 - e.side_length_[2];
            const double dis = std::sqrt(ss_ref(d0, d1, d2));
            const bool hit = dis < thr;
            T.visits++;
            const double mar = std::fabs(dis - thr);
            if (mar < T.min_margin) { T.min_margin = mar; T.min_margin_ulps = mar / u; }
            if (mar <= 64 * u) {
              T.near_calls++;
              // (only a near call can flip: the variants move dis and the threshold by a few ulps at most)
              for (int k = 0; k < 3; k++) T.match_flips[k] += (std::sqrt(ss_var(k, d0, d1, d2)) < thr_v[k]) != hit;
              T.thr_ulp_flips[0] += (dis < thr_up) != hit;
              T.thr_ulp_flips[1] += (dis < thr_dn) != hit;
            }
          }
        }
  }
  for (const orc_audit &p : part) {
    A->gate_tests += p.gate_tests; A->visits += p.visits; A->near_calls += p.near_calls;
    for (int k = 0; k < 3; k++) { A->gate_flips[k] += p.gate_flips[k]; A->gate_flip_visits[k] += p.gate_flip_visits[k]; A->match_flips[k] += p.match_flips[k]; }
    A->thr_ulp_flips[0] += p.thr_ulp_flips[0]; A->thr_ulp_flips[1] += p.thr_ulp_flips[1];
    if (p.min_margin < A->min_margin) { A->min_margin = p.min_margin; A->min_margin_ulps = p.min_margin_ulps; }
    lower(A->min_gate_margin, p.min_gate_margin);
  }
}

}  // extern "C"

// ---- candidate_verify with two solutions of every hypothesis side by side (sgtd_oracle.h: orc_verify_audit)
namespace {
struct CandPairs {
  std::vector<std::pair<const Desc *, const Desc *>> ml;
  int skip_len = 1, use_size = 0;
};
CandPairs pairs_of(orc_manager *m, int cand) {
  CandPairs c;
  const int64_t lo = m->cand_off[cand], hi = m->cand_off[cand + 1];
  const int64_t n = hi - lo;
  c.ml.resize(n);
  for (int64_t k = 0; k < n; k++) {
    const auto &loc = m->entry_loc[m->cm_e[lo + k]];
    c.ml[k] = {&m->last_query[m->cm_q[lo + k]], &m->data_base_.at(loc.first)[loc.second]};
  }
  c.skip_len = (int)(n / 50) + 1;
  c.use_size = (int)(n / c.skip_len);
  return c;
}
// the three vertex distances of a pair under (rot, t), as pair_inlier computes them (no early exit)
inline void vertex_dists(const Desc &qa, const Desc &db, const double t[3], const M3 &rot, double d[3]) {
  const V3 *qs[3] = {&qa.vertex_A_, &qa.vertex_B_, &qa.vertex_C_};
  const V3 *ds[3] = {&db.vertex_A_, &db.vertex_B_, &db.vertex_C_};
  for (int v = 0; v < 3; v++) {
    V3 p;
    for (int r = 0; r < 3; r++)
      p[r] = (rot.m[r][0] * (*qs[v])[0] + rot.m[r][1] * (*qs[v])[1] + rot.m[r][2] * (*qs[v])[2]) + t[r];
    d[v] = norm3(sub(p, *ds[v]));
  }
}
}  // namespace

int orc_verify_hyp_inputs(orc_manager *m, int cand, double *cov_out, double *qc, double *ec) {
  const CandPairs c = pairs_of(m, cand);
  if (!cov_out) return c.use_size;
  for (int i = 0; i < c.use_size; i++) {
    const Desc &qa = *c.ml[(size_t)i * c.skip_len].first, &db = *c.ml[(size_t)i * c.skip_len].second;
    M3 src{}, ref{};
    const V3 *qs[3] = {&qa.vertex_A_, &qa.vertex_B_, &qa.vertex_C_};
    const V3 *ds[3] = {&db.vertex_A_, &db.vertex_B_, &db.vertex_C_};
    for (int cc = 0; cc < 3; cc++)
      for (int r = 0; r < 3; r++) {
        src.m[r][cc] = (*qs[cc])[r] - qa.center_[r];
        ref.m[r][cc] = (*ds[cc])[r] - db.center_[r];
      }
    const M3 cov = mul(src, transpose(ref));
    for (int r = 0; r < 3; r++) {
      for (int cc = 0; cc < 3; cc++) cov_out[(size_t)i * 9 + r * 3 + cc] = cov.m[r][cc];
      qc[(size_t)i * 3 + r] = qa.center_[r];
      ec[(size_t)i * 3 + r] = db.center_[r];
    }
  }
  return c.use_size;
}

void orc_verify_hyp_solutions(orc_manager *m, int cand, double *rt) {
  const CandPairs c = pairs_of(m, cand);
  for (int i = 0; i < c.use_size; i++) {
    double t[3];
    M3 rot;
    solve_triangle(*c.ml[(size_t)i * c.skip_len].first, *c.ml[(size_t)i * c.skip_len].second, t, rot);
    for (int r = 0; r < 3; r++) {
      for (int cc = 0; cc < 3; cc++) rt[(size_t)i * 12 + r * 3 + cc] = rot.m[r][cc];
      rt[(size_t)i * 12 + 9 + r] = t[r];
    }
  }
}

void orc_audit_verify(orc_manager *m, int cand, const double *other_rt, int n_hyp, orc_verify_audit *A) {
  const CandPairs c = pairs_of(m, cand);
  if (n_hyp != c.use_size) return;
  const int64_t n = (int64_t)c.ml.size();
  const double thr = 3.0;
  A->candidates++;
  std::vector<int> votes_a(c.use_size, 0), votes_b(c.use_size, 0);
  std::vector<M3> ra(c.use_size), rb(c.use_size);
  std::vector<std::array<double, 3>> ta(c.use_size), tb(c.use_size);
  for (int i = 0; i < c.use_size; i++) {
    solve_triangle(*c.ml[(size_t)i * c.skip_len].first, *c.ml[(size_t)i * c.skip_len].second, ta[i].data(), ra[i]);
    for (int r = 0; r < 3; r++) {
      for (int cc = 0; cc < 3; cc++) {
        rb[i].m[r][cc] = other_rt[(size_t)i * 12 + r * 3 + cc];
        A->max_rot_diff = std::max(A->max_rot_diff, std::fabs(rb[i].m[r][cc] - ra[i].m[r][cc]));
      }
      tb[i][r] = other_rt[(size_t)i * 12 + 9 + r];
      A->max_t_diff = std::max(A->max_t_diff, std::fabs(tb[i][r] - ta[i][r]));
    }
    A->hypotheses++;
    for (int64_t j = 0; j < n; j++) {
      double da[3], db_[3];
      vertex_dists(*c.ml[j].first, *c.ml[j].second, ta[i].data(), ra[i], da);
      vertex_dists(*c.ml[j].first, *c.ml[j].second, tb[i].data(), rb[i], db_);
      bool in_a = true, in_b = true;
      for (int v = 0; v < 3; v++) {
        const bool a = da[v] < thr, b = db_[v] < thr;
        in_a = in_a && a; in_b = in_b && b;
        A->vertex_tests++;
        if (a != b) A->vertex_flips++;
        const double mg = std::fabs(da[v] - thr);
        if (mg <= 1e-9) A->near_calls++;
        if (mg < A->min_margin) A->min_margin = mg;
        A->max_norm_diff = std::max(A->max_norm_diff, std::fabs(da[v] - db_[v]));
      }
      A->pair_tests++;
      if (in_a != in_b) A->pair_flips++;
      votes_a[i] += in_a ? 1 : 0;
      votes_b[i] += in_b ? 1 : 0;
    }
    if (votes_a[i] != votes_b[i]) A->vote_list_diffs++;
  }
  auto best_of = [&](const std::vector<int> &vl, int &idx, int &mx) {
    idx = 0; mx = 0;
    for (int i = 0; i < c.use_size; i++)
      if (mx < vl[i]) { idx = i; mx = vl[i]; }
  };
  int ia, ma, ib, mb;
  best_of(votes_a, ia, ma);
  best_of(votes_b, ib, mb);
  if (ia != ib) A->best_index_diffs++;
  // the score (:539) and the set of kept pairs (:516-539) under each solution's own best hypothesis
  auto kept = [&](int mx, const M3 &rot, const double *t, std::vector<char> &flags) {
    flags.assign((size_t)n, 0);
    if (mx < 4) return -1;  // :515
    int ns = 0;
    for (int64_t j = 0; j < n; j++)
      if (pair_inlier(*c.ml[j].first, *c.ml[j].second, t, rot, thr)) { flags[(size_t)j] = 1; ns++; }
    return ns;
  };
  std::vector<char> fa, fb;
  const int sa = c.use_size ? kept(ma, ra[ia], ta[ia].data(), fa) : -1;
  const int sb = c.use_size ? kept(mb, rb[ib], tb[ib].data(), fb) : -1;
  if (sa != sb) A->score_diffs++;
  if (fa != fb) A->inlier_set_diffs++;
}
