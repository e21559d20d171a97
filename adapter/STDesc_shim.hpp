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
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#ifndef SGTD_ACCEL_H
#include "sgtd_accel.h"   // include/sgtd_accel.h of this repository (add its directory to the include path)
#endif

#ifndef SGTD_SHIM_FILL_THREADS
// host threads that fill LOOP_RESULT::loop_std_pair of one SearchLoop call (never more than the host has).  155 000 pairs in 50
// lists on a 256-thread host (tools/fill_bench.cpp): one thread 20 ms; threads started and joined per call 3.4 / 2.6 / 2.4 / 3.3 ms
// with 12 / 24 / 48 / 96 of them — starting and joining them alone costs 0.4 / 0.7 / 1.4 / 2.6 ms; the team below, which sleeps
// between calls, 2.0 / 1.8 / 1.1 / 1.4 ms
#define SGTD_SHIM_FILL_THREADS 48u
#endif

namespace sgtd_shim {

// ---- point-cloud access: pcl::PointCloud<pcl::PointXYZL>::Ptr (->points) or a plain vector
template <class Ptr>
auto points_of(const Ptr &pc) -> decltype((pc->points)) { return pc->points; }
template <class P>
const std::vector<P> &points_of(const std::vector<P> &pc) { return pc; }

// ---- STDesc <-> sgtd_desc_soa ----------------------------------------------------------
struct SoaBuf {
  // one block, NOT value-initialised (a std::vector would zero a megabyte per frame that is overwritten at once)
  std::unique_ptr<unsigned char[]> block;
  size_t n_;
  sgtd_desc_soa v;
  explicit SoaBuf(size_t n) : block(new unsigned char[n * 136 + 64]), n_(n) {
    unsigned char *p = block.get();
    p += (8 - reinterpret_cast<uintptr_t>(p) % 8) % 8;
    v.side = reinterpret_cast<double *>(p); v.angle = v.side + 3 * n; v.center = v.angle + 3 * n;
    v.vertex = reinterpret_cast<float *>(v.center + 3 * n); v.label = reinterpret_cast<int32_t *>(v.vertex + 9 * n);
    v.frame = reinterpret_cast<uint32_t *>(v.label + 3 * n); v.node_id = reinterpret_cast<int32_t *>(v.frame + n);
  }
  size_t capacity() const { return n_; }
};

// The same arrays in page-locked memory (sgtd_host_alloc), kept by the calling thread from call to call
// and grown when a call needs more: the entries SearchLoop fetches (20 MB per frame) are written in place by the
// device (sgtd_search_frame: every array it is handed page-locked -> no copy, one wait) instead of arriving through
// the runtime's pageable staging.  q_idx: the query side of the fetched pairs, kept like the entries.  Falls back
// to ordinary memory if the allocation fails.
struct PinnedSoa {
  sgtd_desc_soa v{};
  int32_t *q_idx = nullptr;
  std::vector<int32_t> plain_q;
  size_t cap = 0;
  bool pinned = false;
  std::vector<void *> owned;
  SoaBuf *plain = nullptr;
  bool reserve(size_t n) {
    if (n <= cap) return true;
    release();
    const size_t want = n + n / 4 + 1024;
    void *p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t bytes[8] = {want * 3 * sizeof(double), want * 3 * sizeof(double), want * 3 * sizeof(double), want * 9 * sizeof(float),
                             want * 3 * sizeof(int32_t), want * sizeof(uint32_t), want * 3 * sizeof(int32_t), want * sizeof(int32_t)};
    bool ok = true;
    for (int i = 0; i < 8 && ok; i++) ok = sgtd_host_alloc(bytes[i], &p[i]) == SGTD_OK && p[i];
    if (ok) {
      owned.assign(p, p + 8);
      v.side = (double *)p[0]; v.angle = (double *)p[1]; v.center = (double *)p[2]; v.vertex = (float *)p[3];
      v.label = (int32_t *)p[4]; v.frame = (uint32_t *)p[5]; v.node_id = (int32_t *)p[6];
      q_idx = (int32_t *)p[7];
      pinned = true;
    } else {
      for (void *q : p) if (q) sgtd_host_free(q);
      plain = new SoaBuf(want);
      v = plain->v;
      plain_q.resize(want);
      q_idx = plain_q.data();
      pinned = false;
    }
    cap = want;
    return true;
  }
  void release() {
    for (void *q : owned) sgtd_host_free(q);
    owned.clear();
    delete plain; plain = nullptr;
    cap = 0; v = sgtd_desc_soa{}; q_idx = nullptr;
  }
  PinnedSoa() = default;
  PinnedSoa(const PinnedSoa &) = delete;
  PinnedSoa &operator=(const PinnedSoa &) = delete;
  // (a thread that ends — the process's main thread at exit — does not call into the HIP runtime any
  // more: page-locked buffers are given back by release_thread_buffers(), otherwise with the process)
  ~PinnedSoa() { delete plain; }
};
inline PinnedSoa &fetched_entries() { static thread_local PinnedSoa p; return p; }
// Gives the calling thread's page-locked buffers back.  MANDATORY before a thread that has called SearchLoop or
// candidate_selector ends (its thread_local destructor no longer calls into the HIP runtime: the eight blocks, some
// 130 B per fetched entry, would stay page-locked until the process exits); ~STDescManager calls it for the thread
// that destroys the manager.  The reference calls every method from its main thread (not re-entrant).
inline void release_thread_buffers() { fetched_entries().release(); }

template <class Desc>
void to_soa(const std::vector<Desc> &in, SoaBuf &b) {
  for (size_t i = 0; i < in.size(); i++) {
    for (int k = 0; k < 3; k++) {
      b.v.side[3 * i + k] = in[i].side_length_[k]; b.v.angle[3 * i + k] = in[i].angle_[k]; b.v.center[3 * i + k] = in[i].center_[k];
      b.v.vertex[9 * i + k] = (float)in[i].vertex_A_[k]; b.v.vertex[9 * i + 3 + k] = (float)in[i].vertex_B_[k];
      b.v.vertex[9 * i + 6 + k] = (float)in[i].vertex_C_[k];
      b.v.label[3 * i + k] = (int32_t)in[i].vertex_attached_[k];          // (int) as STDesc.cpp:158-160
      b.v.node_id[3 * i + k] = in[i].node_id.size() == 3 ? in[i].node_id[k] : 0;
    }
    b.v.frame[i] = in[i].frame_id_;
  }
}

// one STDesc from entry i of a structure of arrays
template <class Desc>
inline Desc desc_from(const sgtd_desc_soa &e, size_t i) {
  Desc d;
  for (int c = 0; c < 3; c++) {
    d.side_length_[c] = e.side[3 * i + c]; d.angle_[c] = e.angle[3 * i + c]; d.center_[c] = e.center[3 * i + c];
    d.vertex_A_[c] = e.vertex[9 * i + c]; d.vertex_B_[c] = e.vertex[9 * i + 3 + c]; d.vertex_C_[c] = e.vertex[9 * i + 6 + c];
    d.vertex_attached_[c] = (double)e.label[3 * i + c];
  }
  d.frame_id_ = e.frame[i];
  d.node_id = {e.node_id[3 * i], e.node_id[3 * i + 1], e.node_id[3 * i + 2]};
  return d;
}

// (a frame's few thousand descriptors, constructed in place by the calling thread)
template <class Desc>
void from_soa(const SoaBuf &b, size_t n, std::vector<Desc> &out) {
  out.clear();
  out.reserve(n);
  for (size_t i = 0; i < n; i++) out.push_back(desc_from<Desc>(b.v, i));
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
// the device part of candidate_selector: candidates, votes, list offsets and the (query
// descriptor, table entry) index pairs of every list, in the reference's order
// fill(k0, k1) builds the lists k0 .. k1 - 1; the n lists (off: n + 1 offsets into the pairs) are handed one by one, in their
// order (candidates come by votes: the long lists first), to a team of up to SGTD_SHIM_FILL_THREADS threads, the caller among
// them.  The team is started at the first call that needs it and SLEEPS on a condition variable between calls: threads started
// and joined per call cost more than they filled (above), and an OpenMP team, which spins after its region, slowed the HIP calls
// that follow (8 threads: 38.7 ms per frame against 27.9 single-threaded).  A fill that throws (std::bad_alloc on a match list
// of 10^5 pairs) is rethrown on the caller once every list that was started has finished.  Callers on several threads take
// turns.  The team's threads end with the process (the function-local static's destructor wakes and joins them).
class FillTeam {
  std::mutex m_, turn_;
  std::condition_variable work_, done_;
  std::vector<std::thread> th_;
  void (*call_)(void *, int) = nullptr;
  void *ctx_ = nullptr;
  int n_jobs_ = 0, next_ = 0, running_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
  std::exception_ptr failed_;
  void take(std::unique_lock<std::mutex> &l) {      // (m_ held) run jobs until none is left to start
    while (next_ < n_jobs_) {
      const int j = next_++;
      running_++;
      l.unlock();
      std::exception_ptr ex;
      try { call_(ctx_, j); } catch (...) { ex = std::current_exception(); }
      l.lock();
      running_--;
      if (ex) { if (!failed_) failed_ = ex; next_ = n_jobs_; }      // (nothing new is started after a failure)
    }
  }
  void loop() {
    unsigned long seen = 0;
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      work_.wait(l, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      take(l);
      if (running_ == 0) done_.notify_all();
    }
  }

 public:
  FillTeam() = default;
  FillTeam(const FillTeam &) = delete;
  FillTeam &operator=(const FillTeam &) = delete;
  ~FillTeam() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    work_.notify_all();
    for (auto &t : th_) if (t.joinable()) t.join();
  }
  // f(j) for j = 0 .. n - 1 on up to `threads` threads (the caller included); returns when all have finished
  template <class F>
  void run(int n, unsigned threads, F &f) {
    std::lock_guard<std::mutex> turn(turn_);
    std::unique_lock<std::mutex> l(m_);
    // (a thread that cannot be started — std::system_error — leaves a smaller team: the caller fills what nobody takes)
    try { while (th_.size() + 1 < threads) th_.emplace_back([this] { loop(); }); } catch (const std::system_error &) {}
    call_ = [](void *c, int j) { (*static_cast<F *>(c))(j); };
    ctx_ = &f; n_jobs_ = n; next_ = 0; failed_ = nullptr; gen_++;
    work_.notify_all();
    take(l);
    done_.wait(l, [&] { return running_ == 0; });
    n_jobs_ = 0; ctx_ = nullptr;
    if (failed_) { std::exception_ptr ex = failed_; failed_ = nullptr; std::rethrow_exception(ex); }
  }
};
inline FillTeam &fill_team() { static FillTeam t; return t; }

template <class F>
void deal_lists(int n, const int64_t *off, F &&fill) {
  const int64_t total = off[n] - off[0];
  const unsigned n_thr = total > 8192 ? std::min<unsigned>(SGTD_SHIM_FILL_THREADS, std::max(1u, std::thread::hardware_concurrency())) : 1u;
  if (n_thr <= 1 || n <= 1) { fill(0, n); return; }
  auto one = [&](int k) { fill(k, k + 1); };
  fill_team().run(n, n_thr, one);
}

// Which candidates' LOOP_RESULT::loop_std_pair SearchLoop builds.  0 (default): every candidate's, as the reference does
// (STDesc.cpp:105-131) — 155 000 pair<STDesc, STDesc> per frame on a 10 000-frame map, 3-4.5 ms of host time, which is
// most of the call.  1 (SGTD_SHIM_FILL=best in the environment, or fill_policy() = 1): only the best candidate's, which
// is what SearchLoop hands back as loop_std_pair; the other candidates keep match_id, match_fitness and loop_transform
// and an EMPTY list.  A deviation: the node copies the list of whichever candidate its GICP step prefers
// (semantic_graph_localization.cpp:709,718) and only draws it (:787-788, :858) — with policy 1 that drawing is empty
// unless GICP prefers the best-scored candidate.  The maintainer's choice; never the default.
inline int &fill_policy() {
  static int p = [] { const char *o = std::getenv("SGTD_SHIM_FILL"); return (o && o[0] == 'b') ? 1 : 0; }();
  return p;
}

struct Selection {
  int32_t n_cand = 0;
  std::vector<int32_t> frame, votes, q_idx;
  std::vector<int64_t> off, entry;
};

// with_pairs = false: candidates, votes and list offsets only (SearchLoop verifies on the device and
// fetches just the inlier pairs)
template <class Desc>
int select(sgtd_handle h, const std::vector<Desc> &stds_vec, int candidate_num, Selection &s, bool with_pairs = true) {
  SoaBuf q(stds_vec.size());
  to_soa(stds_vec, q);
  int st = sgtd_query_descs(h, &q.v, (int64_t)stds_vec.size());
  if (st != SGTD_OK) return st;
  const int cn = candidate_num;
  s.frame.assign(cn, -1); s.votes.assign(cn, 0); s.off.assign(cn + 1, 0);
  st = sgtd_result_candidates(h, &s.n_cand, s.frame.data(), s.votes.data(), s.off.data());
  if (st != SGTD_OK || !with_pairs) return st;
  const int64_t total = s.off[s.n_cand];
  s.q_idx.resize((size_t)total); s.entry.resize((size_t)total);
  int64_t got = 0;
  return sgtd_result_pairs(h, 0, s.q_idx.data(), s.entry.data(), total, &got);
}

template <class Desc, class MatchList>
int candidate_selector(sgtd_handle h, const std::vector<Desc> &stds_vec, std::vector<MatchList> &candidate_matcher_vec,
                       unsigned int current_frame_id, int candidate_num, int &CS1) {
  const auto t1 = std::chrono::high_resolution_clock::now();
  Selection s;
  PinnedSoa &pe = fetched_entries();               // the table side of every pair<STDesc, STDesc> and its query index: page-locked, reused
  const int32_t *qip = nullptr;                    // query descriptor of pair r
  int64_t total = 0;
  // ONE call and one wait: the selection, and every pair of every match list with the table entry it names written in place by the
  // device (sgtd_search_frame, SGTD_FRAME_LISTS_ONLY) — sgtd_query_descs + sgtd_result_candidates + sgtd_result_pairs +
  // sgtd_fetch_entries before: four waits, the entry ids down and up again, seven transfers for the entries
  int st = SGTD_ERR_UNSUPPORTED;
  if (!stds_vec.empty()) {
    const int cn = candidate_num;
    SoaBuf qb(stds_vec.size());
    to_soa(stds_vec, qb);
    s.frame.assign(cn, -1); s.votes.assign(cn, 0); s.off.assign((size_t)cn + 1, 0);
    pe.reserve(16384);
    for (int attempt = 0; attempt < 4; attempt++) {
      sgtd_frame_search fs{};
      fs.flags = SGTD_FRAME_LISTS_ONLY;
      fs.cand_frame = s.frame.data(); fs.cand_votes = s.votes.data(); fs.pair_off = s.off.data();
      fs.inlier_q_idx = pe.q_idx; fs.entries = pe.v; fs.capacity = (int64_t)pe.cap;
      st = sgtd_search_frame(h, &qb.v, (int64_t)stds_vec.size(), &fs);
      s.n_cand = fs.n_cand;
      total = fs.n_inliers;
      if (st != SGTD_ERR_CAPACITY) break;
      pe.reserve((size_t)total + (size_t)total / 4);     // (a first frame, a frame with longer lists than any before: once more, with room)
    }
    qip = pe.q_idx;
  }
  if (st == SGTD_ERR_UNSUPPORTED) {                // (an empty frame; a handle over several devices: the calls one after the other)
    st = select(h, stds_vec, candidate_num, s);
    if (st != SGTD_OK) return st;
    total = s.off[s.n_cand];
    pe.reserve((size_t)total);
    st = sgtd_fetch_entries(h, s.entry.data(), total, &pe.v);
    qip = s.q_idx.data();
  }
  if (st != SGTD_OK) return st;
  const sgtd_desc_soa &ent = pe.v;
  const size_t first = candidate_matcher_vec.size();
  candidate_matcher_vec.resize(first + (size_t)s.n_cand);
  // every pair constructed in place (no resize: it would zero 832 bytes per pair first), the lists dealt to threads
  deal_lists(s.n_cand, s.off.data(), [&](int k0, int k1) {
    for (int k = k0; k < k1; k++) {
      MatchList &ml = candidate_matcher_vec[first + (size_t)k];
      ml.match_id_.first = (int)current_frame_id;    // :436
      ml.match_id_.second = s.frame[k];              // :437
      ml.match_list_.reserve((size_t)(s.off[k + 1] - s.off[k]));
      for (int64_t r = s.off[k]; r < s.off[k + 1]; r++)
        ml.match_list_.emplace_back(stds_vec[(size_t)qip[(size_t)r]], desc_from<Desc>(ent, (size_t)r));
    }
  });
  const auto t2 = std::chrono::high_resolution_clock::now();
  CS1 = (int)(std::chrono::duration<double>(t2 - t1).count() * 1000);   // int CS1 truncates a double ms value, :455
  return SGTD_OK;
}

// milliseconds SearchLoop spent, by part, summed over this thread's calls: the device work and its
// transfers (select .. inliers) and the host-side construction of the reference's result containers (fill)
struct SearchTiming { double select = 0, verify = 0, inliers = 0, fill = 0; long calls = 0; };   // inliers: the inlier pairs and their table entries
inline SearchTiming &search_timing() { static thread_local SearchTiming t; return t; }

// ---- STDesc.cpp:84-147 with candidate_verify (:462-547) on the device ---------------------
// Vec3 / Mat3 = Eigen::Vector3d / Eigen::Matrix3d (operator[] and operator()(row, col)).
// The match lists stay on the device: only the inlier pairs of every candidate
// (sucess_match_vec, what LOOP_RESULT::loop_std_pair holds) are fetched — one device compaction and
// one gather of the table entries they name, back to back (sgtd_result_inlier_entries) — instead of
// the ~10^5 pairs of the full lists.
template <class Desc, class Vec3, class Mat3, class LoopResult>
int SearchLoop(sgtd_handle h, const std::vector<Desc> &stds_vec, std::pair<int, double> &loop_result,
               std::pair<Vec3, Mat3> &loop_transform, std::vector<std::pair<Desc, Desc>> &loop_std_pair,
               std::vector<LoopResult> &match_result_list, unsigned int current_frame_id, int candidate_num,
               double icp_threshold, int &CS1) {
  (void)current_frame_id;
  loop_result = std::pair<int, double>(-1, 0);
  if (stds_vec.empty()) return SGTD_OK;            // "No STDescs!" (:89-93)
  const auto t1 = std::chrono::high_resolution_clock::now();
  // where the call's time went (search_timing(), summed over the calls of this thread)
  auto last = t1;
  auto lap = [&](double SearchTiming::*part, const char *what) {
    const auto now = std::chrono::high_resolution_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - last).count();
    search_timing().*part += ms;
    last = now;
#ifdef SGTD_SHIM_TIMING
    std::fprintf(stderr, "  [shim] %-10s %.3f ms\n", what, ms);
#else
    (void)what;
#endif
  };
#define SGTD_LAP(part) lap(&SearchTiming::part, #part)
  Selection s;
  const int cn = candidate_num;
  std::vector<double> score(cn), pose((size_t)cn * 12);   // rot row-major (9), then t (3)
  std::vector<int64_t> ioff((size_t)cn + 1, 0);
  int64_t n_inl = 0;
  PinnedSoa &pe = fetched_entries();      // page-locked, reused from frame to frame
  // ONE call for candidate_selector (:98), candidate_verify of every candidate (:105-118) and the inlier pairs of every
  // candidate (sucess_match_vec, :516-539) with the table entries they name: one wait for the device instead of eight
  // (sgtd_search_frame).  Room for the inlier pairs: what the frames before needed, with slack; a frame that needs more
  // says so and its pairs are fetched by the second call below.
  int st;
  {
    SoaBuf qb(stds_vec.size());
    to_soa(stds_vec, qb);
#ifdef SGTD_SHIM_TIMING
    std::fprintf(stderr, "  [shim] to_soa done at %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t1).count());
#endif
    s.frame.assign(cn, -1); s.votes.assign(cn, 0); s.off.assign((size_t)cn + 1, 0);
    pe.reserve(16384);
    sgtd_frame_search fs{};
    fs.cand_frame = s.frame.data(); fs.cand_votes = s.votes.data(); fs.pair_off = s.off.data();
    fs.score = score.data(); fs.pose = pose.data(); fs.inlier_off = ioff.data();
    fs.inlier_q_idx = pe.q_idx; fs.entries = pe.v; fs.capacity = (int64_t)pe.cap;
    st = sgtd_search_frame(h, &qb.v, (int64_t)stds_vec.size(), &fs);
#ifdef SGTD_SHIM_TIMING
    std::fprintf(stderr, "  [shim] sgtd_search_frame back at %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t1).count());
#endif
    s.n_cand = fs.n_cand;
    n_inl = fs.n_inliers;
    if (st == SGTD_ERR_CAPACITY) {         // more inlier pairs than there was room for: everything else is there
      pe.reserve((size_t)n_inl + (size_t)n_inl / 2);
      st = sgtd_result_inlier_entries(h, 0, ioff.data(), pe.q_idx, &pe.v, (int64_t)pe.cap, &n_inl);
    }
  }
  if (st == SGTD_ERR_UNSUPPORTED) {        // (a handle over several devices: the calls one after the other)
    st = select(h, stds_vec, candidate_num, s, /*with_pairs=*/false);  // :98
    if (st != SGTD_OK) return st;
    CS1 = (int)(std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t1).count() * 1000);
    SGTD_LAP(select);
    st = sgtd_verify(h);                             // :105-118 for every candidate
    if (st != SGTD_OK) return st;
    st = sgtd_result_verify(h, 0, score.data(), pose.data());
    if (st != SGTD_OK) return st;
    SGTD_LAP(verify);
    const int64_t most = s.off[s.n_cand];   // every pair of every candidate's list
    pe.reserve((size_t)most);
    st = sgtd_result_inlier_entries(h, 0, ioff.data(), pe.q_idx, &pe.v, most, &n_inl);
    if (st != SGTD_OK) return st;
    SGTD_LAP(inliers);
  } else {
    if (st != SGTD_OK) return st;
    CS1 = (int)(std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t1).count() * 1000);
    SGTD_LAP(select);                        // (the whole device side of the call)
  }
  const sgtd_desc_soa &ent = pe.v;
  const int32_t *const iqp = pe.q_idx;       // (a thread_local's: the fill threads below must see THIS thread's)
#ifdef SGTD_SHIM_TIMING
  std::fprintf(stderr, "  [shim] %lld inlier pairs over %d candidates\n", (long long)n_inl, s.n_cand);
#endif
  double best_score = 0;
  int best = -1;
  const size_t first = match_result_list.size();
  match_result_list.resize(first + (size_t)s.n_cand);
  for (int k = 0; k < s.n_cand; k++) {             // :105-131
    LoopResult &r = match_result_list[first + (size_t)k];
    r.match_id = s.frame[k];
    r.match_fitness = score[k];                    // an int member in the reference: truncates like :119
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) r.loop_transform.second(a, b) = pose[(size_t)k * 12 + a * 3 + b];
      r.loop_transform.first[a] = pose[(size_t)k * 12 + 9 + a];
    }
    // (a candidate that failed verification — score -1 — has no sucess_match_vec; the others' are built by
    // the fill threads below)
    if (score[k] > best_score) { best_score = score[k]; best = k; }   // :125-131
  }
  // LOOP_RESULT::loop_std_pair of EVERY candidate is filled: the caller copies the list of whichever
  // candidate its registration step prefers (semantic_graph_localization.cpp:622,672-718), and the
  // member is a plain std::vector — nothing can be deferred.
  // The ~10^5 descriptors (416 bytes and a heap-allocated node_id each) are written by a team of
  // threads, every candidate's list by one of them (deal_lists).  Filling the query side of
  // the pairs while the table side is still being fetched was measured too: the fetch then takes 18-22 ms
  // instead of 3-4 (the copy's own host threads lose their cores) — the fill starts after it.
  // Every pair is CONSTRUCTED in place (reserve + emplace_back: the query descriptor copied, the table
  // descriptor built and moved), not value-initialised by a resize and then assigned: a resize
  // zeroes the 832 bytes of every pair first — 129 MB per frame, on the calling thread.
  const int only = fill_policy() == 1 ? best : -2;
  deal_lists(s.n_cand, ioff.data(), [&](int k0, int k1) {
    for (int k = k0; k < k1; k++) {
      if (!(score[k] >= 0) || (only != -2 && k != only)) continue;
      std::vector<std::pair<Desc, Desc>> &lp = match_result_list[first + (size_t)k].loop_std_pair;
      lp.reserve((size_t)(ioff[(size_t)k + 1] - ioff[(size_t)k]));
      for (int64_t j = ioff[(size_t)k]; j < ioff[(size_t)k + 1]; j++)
        lp.emplace_back(stds_vec[(size_t)iqp[(size_t)j]], desc_from<Desc>(ent, (size_t)j));
    }
  });
  SGTD_LAP(fill);
#undef SGTD_LAP
  search_timing().calls++;
  if (best >= 0 && best_score > icp_threshold) {   // :138-146
    const LoopResult &b = match_result_list[first + (size_t)best];
    loop_result = std::pair<int, double>(b.match_id, best_score);
    loop_transform = b.loop_transform;
    loop_std_pair = b.loop_std_pair;
  }
  return SGTD_OK;
}

}  // namespace sgtd_shim
