// fill_bench.cpp — the host-side construction of LOOP_RESULT::loop_std_pair (adapter/STDesc_shim.hpp, SearchLoop's last part) on
// stand-in data of a 10 000-frame map's size: 50 lists, 155 000 pair<STDesc, STDesc>, by short-lived threads (what the adapter did
// up to round 6) and by a team that sleeps between calls.  Host only:  g++ -O2 -std=c++17 -Iinclude -Iadapter -pthread tools/fill_bench.cpp
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <malloc.h>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
#include "sgtd/STDescManager.hpp"
using Desc = sgtd::STDesc;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

class Team {      // n - 1 sleeping threads + the caller
  std::mutex m; std::condition_variable work, done;
  std::vector<std::thread> th;
  const std::function<void(int)> *job = nullptr;
  int n_jobs = 0, next = 0, running = 0; unsigned long gen = 0; bool stop = false;
  void loop() {
    unsigned long seen = 0;
    std::unique_lock<std::mutex> l(m);
    for (;;) {
      work.wait(l, [&] { return stop || gen != seen; });
      if (stop) return;
      seen = gen;
      while (next < n_jobs) { const int j = next++; running++; l.unlock(); (*job)(j); l.lock(); running--; }
      if (running == 0) done.notify_one();
    }
  }
 public:
  explicit Team(int n) { for (int i = 0; i + 1 < n; i++) th.emplace_back([this] { loop(); }); }
  ~Team() { { std::lock_guard<std::mutex> l(m); stop = true; } work.notify_all(); for (auto &t : th) t.join(); }
  void run(int n, const std::function<void(int)> &f) {
    std::unique_lock<std::mutex> l(m);
    job = &f; n_jobs = n; next = 0; gen++;
    work.notify_all();
    while (next < n_jobs) { const int j = next++; running++; l.unlock(); f(j); l.lock(); running--; }
    done.wait(l, [&] { return running == 0 && next >= n_jobs; });
  }
};

int main(int argc, char **argv) {
  if (!getenv("FILL_DEFAULT_MALLOC")) { mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 256 << 20); }      // (as examples/localize.cpp)
  const int n_thr = argc > 1 ? atoi(argv[1]) : 12, reps = argc > 2 ? atoi(argv[2]) : 20;
  const int cn = 50; const size_t per = 3100, total = cn * per, nq = 7000;
  std::mt19937 rng(7);
  std::vector<Desc> stds(nq);
  for (auto &d : stds) { d.node_id = {1, 2, 3}; d.frame_id_ = 5; }
  std::vector<double> side(total * 3, 1.5), angle(total * 3, 0.5), center(total * 3, 2.5);
  std::vector<float> vertex(total * 9, 3.f);
  std::vector<int> label(total * 3, 4), node(total * 3, 9), iq(total);
  std::vector<unsigned> frame(total, 77);
  for (auto &q : iq) q = (int)(rng() % nq);
  auto desc_from = [&](size_t i) {
    Desc d;
    for (int c = 0; c < 3; c++) {
      d.side_length_[c] = side[3 * i + c]; d.angle_[c] = angle[3 * i + c]; d.center_[c] = center[3 * i + c];
      d.vertex_A_[c] = vertex[9 * i + c]; d.vertex_B_[c] = vertex[9 * i + 3 + c]; d.vertex_C_[c] = vertex[9 * i + 6 + c];
      d.vertex_attached_[c] = (double)label[3 * i + c];
    }
    d.frame_id_ = frame[i];
    d.node_id = {node[3 * i], node[3 * i + 1], node[3 * i + 2]};
    return d;
  };
  auto fill_list = [&](std::vector<std::pair<Desc, Desc>> &lp, int k) {
    lp.reserve(per);
    for (size_t j = k * per; j < (k + 1) * per; j++) lp.emplace_back(stds[(size_t)iq[j]], desc_from(j));
  };
  Team team(n_thr);
  double t_chunk = 0, t_free_team = 0, t_spawn = 0, t_team = 0, t_noop_spawn = 0, t_noop_team = 0, t_free = 0, t_one = 0;
  for (int r = 0; r < reps + 2; r++) {
    std::vector<std::vector<std::pair<Desc, Desc>>> a(cn), b(cn), c(cn);
    double t0 = now_ms();
    { std::vector<std::thread> th; for (int t = 0; t < n_thr - 1; t++) th.emplace_back([&, t] { for (int k = t; k < cn; k += n_thr) fill_list(a[k], k); });
      for (int k = n_thr - 1; k < cn; k += n_thr) fill_list(a[k], k);
      for (auto &t : th) t.join(); }
    double t1 = now_ms();
    team.run(cn, [&](int k) { fill_list(b[k], k); });
    double t2 = now_ms();
    // the team in two rounds: every list sized by a thread (value-initialised pairs), then chunks of 1024 pairs assigned — no
    // thread is left with a whole long list
    {
      std::vector<std::vector<std::pair<Desc, Desc>>> e(cn);
      double u0 = now_ms();
      team.run(cn, [&](int k) { e[k].resize(per); });
      const int chunk = 1024, per_list = (int)((per + chunk - 1) / chunk);
      team.run(cn * per_list, [&](int j) {
        const int k = j / per_list; const size_t a = (size_t)(j % per_list) * chunk, b = std::min(per, a + chunk);
        for (size_t i = a; i < b; i++) e[k][i] = std::pair<Desc, Desc>(stds[(size_t)iq[k * per + i]], desc_from(k * per + i));
      });
      if (r >= 2) t_chunk += now_ms() - u0;
      double u1 = now_ms();
      team.run(cn, [&](int k) { std::vector<std::pair<Desc, Desc>>().swap(e[k]); });      // the lists freed by the team as well
      if (r >= 2) t_free_team += now_ms() - u1;
    }
    const double t2b = now_ms();
    { std::vector<std::thread> th; for (int t = 0; t < n_thr - 1; t++) th.emplace_back([] {}); for (auto &t : th) t.join(); }
    double t3 = now_ms();
    team.run(n_thr, [](int) {});
    double t4 = now_ms();
    if (r == 0) { for (int k = 0; k < cn; k++) fill_list(c[k], k); t_one = now_ms() - t4; }
    double t5 = now_ms();
    a.clear(); b.clear();
    double t6 = now_ms();
    if (r >= 2) { t_spawn += t1 - t0; t_team += t2 - t1; t_noop_spawn += t3 - t2b; t_noop_team += t4 - t3; t_free += (t6 - t5) / 2; }
  }
  // BuildSingleScanSTD's result: 7 000 descriptors from the structure of arrays, by the caller and by the team in chunks of 512
  {
    double t_one7 = 0, t_team7 = 0;
    const size_t n7 = 7000;
    for (int r = 0; r < reps + 2; r++) {
      std::vector<Desc> a, b;
      double u0 = now_ms();
      a.reserve(n7);
      for (size_t i = 0; i < n7; i++) a.push_back(desc_from(i));
      double u1 = now_ms();
      b.resize(n7);
      team.run((int)((n7 + 511) / 512), [&](int c) { for (size_t i = (size_t)c * 512; i < std::min(n7, ((size_t)c + 1) * 512); i++) b[i] = desc_from(i); });
      double u2 = now_ms();
      if (r >= 2) { t_one7 += u1 - u0; t_team7 += u2 - u1; }
    }
    std::printf("7000 descriptors out of the arrays: the caller alone %.3f ms, sized by the caller and assigned by the team in chunks of 512 %.3f ms\n", t_one7 / reps, t_team7 / reps);
  }
  // SearchLoop's way in: the frame's 7 000 STDesc into the structure of arrays (adapter: to_soa), by the caller and by the team
  {
    double t_one = 0, t_team8 = 0;
    std::vector<double> so(nq * 9); std::vector<float> vo(nq * 9); std::vector<int> lo_(nq * 6); std::vector<unsigned> fo(nq);
    auto put = [&](size_t i) {
      for (int k = 0; k < 3; k++) {
        so[3 * i + k] = stds[i].side_length_[k]; so[3 * nq + 3 * i + k] = stds[i].angle_[k]; so[6 * nq + 3 * i + k] = stds[i].center_[k];
        vo[9 * i + k] = (float)stds[i].vertex_A_[k]; vo[9 * i + 3 + k] = (float)stds[i].vertex_B_[k]; vo[9 * i + 6 + k] = (float)stds[i].vertex_C_[k];
        lo_[3 * i + k] = (int)stds[i].vertex_attached_[k]; lo_[3 * nq + 3 * i + k] = stds[i].node_id.size() == 3 ? stds[i].node_id[k] : 0;
      }
      fo[i] = stds[i].frame_id_;
    };
    for (int r = 0; r < reps + 2; r++) {
      double u0 = now_ms();
      for (size_t i = 0; i < nq; i++) put(i);
      double u1 = now_ms();
      team.run((int)((nq + 511) / 512), [&](int c) { for (size_t i = (size_t)c * 512; i < std::min(nq, ((size_t)c + 1) * 512); i++) put(i); });
      double u2 = now_ms();
      if (r >= 2) { t_one += u1 - u0; t_team8 += u2 - u1; }
    }
    std::printf("7000 descriptors into the arrays: the caller alone %.3f ms, the team in chunks of 512 %.3f ms\n", t_one / reps, t_team8 / reps);
  }
  std::printf("%d threads, %zu pairs in %d lists: short-lived threads %.3f ms, sleeping team %.3f ms | starting and joining idle threads %.3f ms, waking the idle team %.3f ms | one thread %.3f ms | freeing a frame's lists %.3f ms | team in two rounds (sized by list, assigned in chunks of 1024) %.3f ms, lists freed by the team %.3f ms\n",
              n_thr, total, cn, t_spawn / reps, t_team / reps, t_noop_spawn / reps, t_noop_team / reps, t_one, t_free / reps, t_chunk / reps, t_free_team / reps);
  return 0;
}
