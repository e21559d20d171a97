// shim_driver.cpp — drives adapter/STDesc_shim.hpp (through include/sgtd/STDescManager.hpp, the same templates on plain
// structs) against stub_abi.cpp under the sanitizers: BuildSingleScanSTD / AddSTDescs for a map, then SearchLoop and
// candidate_selector frame after frame — the reference's call pattern (semantic_graph_localization.cpp:590-603) — on
// TWO managers in two threads at once (each thread its own manager and its own thread_local page-locked buffers; the
// fill teams of both run concurrently).  Checks what the adapter must guarantee whatever the library returns: list
// lengths equal the offsets it was given, every pair's query side is a descriptor of the frame, loop_std_pair of the
// winner is the winner's list.
//
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined shim_driver.cpp stub_abi.cpp -I../../../include -pthread
//   g++ -std=c++17 -O1 -g -fsanitize=thread            (the same)
#include <cstdio>
#include <cstdlib>
#include <new>
#include <thread>

#include "sgtd/STDescManager.hpp"

static int run(int seed, int frames, int calls) {
  sgtd::ConfigSetting cfg;
  cfg.candidate_num_ = seed % 2 ? 50 : 7;
  sgtd::STDescManager mgr(cfg);
  uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(seed + 1);
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  auto cloud = [&](int n) {
    std::vector<sgtd::PointXYZL> pc((size_t)n);
    for (auto &p : pc) { p.x = (float)(rnd() % 10000) / 100.f; p.y = (float)(rnd() % 10000) / 100.f; p.z = (float)(rnd() % 300) / 100.f; p.label = 3 + (uint32_t)(rnd() % 9); }
    return pc;
  };
  std::vector<sgtd::STDesc> stds;
  for (int f = 0; f < frames; f++) {
    mgr.BuildSingleScanSTD(cloud(20 + (int)(rnd() % 60)), stds);
    if (mgr.last_status() != SGTD_OK) return 10;
    mgr.AddSTDescs(stds);
    if (mgr.last_status() != SGTD_OK || mgr.current_frame_id_ != (unsigned)f + 1) return 11;
  }
  long pairs = 0, loops = 0;
  for (int c = 0; c < calls; c++) {
    mgr.BuildSingleScanSTD(cloud(c % 7 == 0 ? 0 : 30 + (int)(rnd() % 50)), stds);     // (an empty scan: "No STDescs!")
    std::pair<int, double> loop_result;
    std::pair<sgtd::Vec3, sgtd::Mat3> loop_transform;
    std::vector<std::pair<sgtd::STDesc, sgtd::STDesc>> loop_std_pair;
    std::vector<sgtd::LOOP_RESULT> results;
    mgr.SearchLoop(stds, loop_result, loop_transform, loop_std_pair, results);
    if (mgr.last_status() != SGTD_OK) return 12;
    for (const auto &r : results) {
      pairs += (long)r.loop_std_pair.size();
      for (const auto &p : r.loop_std_pair)
        if (p.first.node_id.size() != 3 || p.second.node_id.size() != 3) return 13;
      // (the stub's score = its inlier count; with SGTD_SHIM_FILL=best only the winner's list is built)
      const bool built = sgtd_shim::fill_policy() == 0 || (loop_result.first >= 0 && r.match_id == loop_result.first && r.match_fitness == (int)loop_result.second);
      if (r.match_fitness >= 0 && built && (long)r.loop_std_pair.size() != r.match_fitness) return 14;
      if (sgtd_shim::fill_policy() == 1 && !built && !r.loop_std_pair.empty()) return 18;
    }
    if (loop_result.first >= 0) {
      loops++;
      if (loop_std_pair.empty()) return 15;
    }
    if (c % 3 == 0) {
      std::vector<sgtd::STDMatchList> lists;
      mgr.candidate_selector(stds, lists);
      if (mgr.last_status() != SGTD_OK) return 16;
      for (const auto &l : lists) {
        pairs += (long)l.match_list_.size();
        if (l.match_id_.first != (int)mgr.current_frame_id_) return 17;
      }
    }
  }
  std::printf("thread %d: %d frames, %d calls, %ld pairs built, %ld loops\n", seed, frames, calls, pairs, loops);
  return 0;
}

// the fill team by itself (sgtd_shim::deal_lists): every list built exactly once by some thread; a fill that throws reaches the
// caller as the exception it threw, after which the team works as before; two callers at once take turns
static int team_checks() {
  const int n = 37;
  std::vector<int64_t> off((size_t)n + 1, 0);
  for (int k = 0; k < n; k++) off[(size_t)k + 1] = off[(size_t)k] + 400 + 50 * (k % 7);      // (above the 8192 pairs below which the caller fills alone)
  auto round = [&](int throw_at) {
    std::vector<std::vector<int64_t>> built((size_t)n);
    std::vector<int> times((size_t)n, 0);
    sgtd_shim::deal_lists(n, off.data(), [&](int k0, int k1) {
      for (int k = k0; k < k1; k++) {
        if (k == throw_at) throw std::bad_alloc();
        times[(size_t)k]++;                                   // (every list is one thread's: no two touch the same element)
        for (int64_t j = off[(size_t)k]; j < off[(size_t)k + 1]; j++) built[(size_t)k].push_back(j * 3);
      }
    });
    for (int k = 0; k < n; k++) {
      if (times[(size_t)k] != 1 || (int64_t)built[(size_t)k].size() != off[(size_t)k + 1] - off[(size_t)k]) return 1;
      if (!built[(size_t)k].empty() && built[(size_t)k].back() != (off[(size_t)k + 1] - 1) * 3) return 2;
    }
    return 0;
  };
  if (int r = round(-1)) return 20 + r;
  for (int at : {0, 11, n - 1}) {
    bool caught = false;
    try { (void)round(at); } catch (const std::bad_alloc &) { caught = true; }
    if (!caught) return 23;
    if (int r = round(-1)) return 24 + r;                     // the team after a failed call
  }
  int rc[2] = {0, 0};
  std::thread a([&] { for (int i = 0; i < 20 && !rc[0]; i++) rc[0] = round(-1); });
  std::thread b([&] { for (int i = 0; i < 20 && !rc[1]; i++) rc[1] = round(-1); });
  a.join(); b.join();
  return rc[0] || rc[1] ? 27 : 0;
}

int main(int argc, char **argv) {
  const int calls = argc > 1 ? std::atoi(argv[1]) : 24;
  if (const int t = team_checks()) { std::printf("FAILED fill team %d\n", t); return 1; }
  int rc[2] = {0, 0};
  std::thread a([&] { rc[0] = run(0, 40, calls); });
  std::thread b([&] { rc[1] = run(1, 25, calls); });
  a.join(); b.join();
  if (rc[0] || rc[1]) { std::printf("FAILED %d %d\n", rc[0], rc[1]); return 1; }
  std::printf("shim under the sanitizers: ok\n");
  return 0;
}
