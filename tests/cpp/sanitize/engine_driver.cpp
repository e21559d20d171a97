// engine_driver.cpp — the engine's host code (sgtd_accel.hip, multi_impl.hip.h) under ASan + UBSan, against hip_stub.cpp.
// No kernel runs: "device" buffers are zeroed host memory and the launch hook below leaves behind what selected kernels
// would have — overflow flags, record needs, pair totals, a frame's packed results — so that the host walks its growth,
// re-run, stale-view, capacity and table-file paths with every copy checked by the sanitizer.  Compiled for the host only
// (hipcc --cuda-host-only: the kernel headers are needed for the argument structs); tests/test_sanitizers.py builds and runs it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <stdint.h>

#include "../../../include/sgtd_accel.h"
// the kernels' argument structs (ProbeBuffers, the frame pack's layout) come with the kernels themselves: in a namespace of
// their own here, so that this file's copies of the host-side launch stubs do not collide with the engine's
namespace kernels_of_the_engine {
#include "../../../sgtd_amd/csrc/build_kernel.hip.h"       // (the engine's own include order: the headers lean on each other)
#include "../../../sgtd_amd/csrc/common.hip.h"
#include "../../../sgtd_amd/csrc/table_kernels.hip.h"
#include "../../../sgtd_amd/csrc/probe_kernels.hip.h"
#include "../../../sgtd_amd/csrc/select_kernels.hip.h"
#include "../../../sgtd_amd/csrc/verify_kernels.hip.h"
}  // namespace kernels_of_the_engine
using kernels_of_the_engine::ProbeBuffers;
using kernels_of_the_engine::frame_pack_bytes;
using kernels_of_the_engine::u32;

extern "C" {
typedef void (*launch_hook_t)(const char *name, void **args, void *user);
void sgtd_stub_set_launch_hook(launch_hook_t h, void *user);
unsigned long long sgtd_stub_launches();
size_t sgtd_stub_device_bytes();
size_t sgtd_stub_device_peak();
size_t sgtd_stub_device_blocks();
size_t sgtd_stub_block_size(const void *p);
}

#define REQUIRE(x) do { if (!(x)) { fprintf(stderr, "engine_driver: %s failed at line %d\n", #x, __LINE__); exit(1); } } while (0)
static sgtd_handle g_last = nullptr;     // (whose error text a failed call reports)
#define OK(call) do { const int st_ = (call); if (st_ != SGTD_OK) { fprintf(stderr, "engine_driver: %s = %d at line %d (%s)\n", #call, st_, __LINE__, g_last ? sgtd_last_error(g_last) : ""); exit(1); } } while (0)

namespace {
struct Scenario {
  int sweep_overflows = 0;          // the next launches of the sweep report that the record buffer was too small ...
  unsigned long long need = 0;      // ... by this many records,
  bool reservations = false;        // ... or that the RESERVATIONS outran it (the host then drops the reservation rate)
  int pair_overflows = 0;           // the next launches of query_base_kernel report that the pair buffer was too small
  unsigned pairs_total = 0;         // candidate pairs of the batch (q_pair_base[nq])
  int cand_num = 50;
  long long frame_inliers = -1;     // pack_frame_kernel: inlier pairs the frame's verification "found" (-1: leave zeros)
  int frame_overflow = 0;           // pack_frame_kernel: the next packs carry a set overflow flag (sgtd_search_frame falls back)
  unsigned long long sweeps = 0, packs = 0;
};

std::vector<const void *> g_gathers_into;      // gather_pair_entries_kernel launches: the `side` array each was handed

void hook(const char *name, void **args, void *user) {
  Scenario &S = *static_cast<Scenario *>(user);
  if (strstr(name, "probe_sorted_kernel")) {
    S.sweeps++;
    const ProbeBuffers &B = *static_cast<const ProbeBuffers *>(args[1]);
    REQUIRE(sgtd_stub_block_size(B.ctr) >= 12 * sizeof(u32));
    // the record buffer the kernel was handed really has the granules it was told (+ the quads read past the last list)
    REQUIRE(sgtd_stub_block_size(B.rec) >= ((size_t)B.rec_cap << SGTD_REC_SHIFT) * sizeof(u32));
    REQUIRE(B.rec_slab >= 1 && B.rec_rate >= 1 && B.rec_rate <= 256);
    if (S.sweep_overflows > 0) {
      S.sweep_overflows--;
      B.overflow()[0] = 1;
      *B.rec_need() = S.need;
      *B.rec_cursor() = S.reservations ? (unsigned long long)B.rec_cap * 4ull : (unsigned long long)B.rec_cap;
    } else {
      *B.rec_cursor() = B.rec_cap / 2;
    }
  } else if (strstr(name, "small_order_kernel")) {
    // a one-frame batch clears its counters inside this kernel (the general form: a memset, which the stand-in performs)
    const kernels_of_the_engine::SmallOrder &O = *static_cast<const kernels_of_the_engine::SmallOrder *>(args[1]);
    REQUIRE(sgtd_stub_block_size(O.ctr) >= (size_t)O.ctr_words * sizeof(u32) && sgtd_stub_block_size(O.pos_of_slot) >= (size_t)O.max_pass_slots * sizeof(u32));
    REQUIRE(!O.votes || sgtd_stub_block_size(O.votes) >= (size_t)O.span * sizeof(u32));
    REQUIRE(sgtd_stub_block_size(O.slot_of_words) >= (size_t)((O.span + 3u) / 4u) * sizeof(u32));
    REQUIRE(sgtd_stub_block_size(O.gid) >= (size_t)O.n_slots * sizeof(u32) && sgtd_stub_block_size(O.order) >= (size_t)O.n_slots * sizeof(u32));
    memset(O.ctr, 0, (size_t)O.ctr_words * sizeof(u32));
  } else if (strstr(name, "head_flags_kernel")) {
    // the bucket count of a segment = (head flag of the last entry) + (exclusive scan at the last entry): with no scan running both
    // reads see this word — half the buckets the table is to "have"
    u32 *flags = *static_cast<u32 **>(args[1]);
    const long long E = *static_cast<long long *>(args[2]);
    REQUIRE(E > 0 && sgtd_stub_block_size(flags) >= (size_t)E * sizeof(u32));
    flags[E - 1] = (u32)(E / 32 + 1);
  } else if (strstr(name, "query_base_kernel")) {
    u32 *q_pair_base = *static_cast<u32 **>(args[1]);
    const int nq = *static_cast<int *>(args[2]);
    int *overflow = *static_cast<int **>(args[4]);
    REQUIRE(sgtd_stub_block_size(q_pair_base) >= (size_t)(nq + 1) * sizeof(u32));
    for (int q = 0; q <= nq; q++) q_pair_base[q] = (u32)((unsigned long long)S.pairs_total * q / nq);
    if (S.pair_overflows > 0) { S.pair_overflows--; overflow[1] = 1; }
  } else if (strstr(name, "block_scan_kernel")) {
    // a one-query batch: the scan leaves the query's base and total itself (no query_base_kernel launch)
    u32 *base_of_one = *static_cast<u32 **>(args[7]);
    int *overflow = *static_cast<int **>(args[6]);
    if (base_of_one) {
      REQUIRE(sgtd_stub_block_size(base_of_one) >= 2 * sizeof(u32));
      base_of_one[0] = 0; base_of_one[1] = S.pairs_total;
      if (S.pair_overflows > 0) { S.pair_overflows--; overflow[1] = 1; }
    }
  } else if (strstr(name, "gather_pair_entries_kernel")) {
    // where the entries go, and that every array the kernel is handed has the room it is told
    const long long cap = *static_cast<long long *>(args[2]);
    int *qi = *static_cast<int **>(args[3]);
    const kernels_of_the_engine::DescArrays &out = *static_cast<const kernels_of_the_engine::DescArrays *>(args[5]);
    auto room_of = [](const void *p) {        // bytes from p to the end of the block it lies in
      hipDeviceptr_t base; size_t size;
      REQUIRE(hipMemGetAddressRange(&base, &size, const_cast<void *>(p)) == hipSuccess);
      return size - (size_t)(static_cast<const char *>(p) - static_cast<const char *>(base));
    };
    REQUIRE(room_of(out.side) >= (size_t)cap * 24 && room_of(out.angle) >= (size_t)cap * 24 && room_of(out.center) >= (size_t)cap * 24);
    REQUIRE(room_of(out.vertex) >= (size_t)cap * 36 && room_of(out.label) >= (size_t)cap * 12 && room_of(out.node_id) >= (size_t)cap * 12);
    REQUIRE(room_of(out.frame) >= (size_t)cap * 4 && room_of(qi) >= (size_t)cap * 4);
    g_gathers_into.push_back(out.side);
  } else if (strstr(name, "pack_frame_kernel")) {
    S.packs++;
    const int cn = *static_cast<int *>(args[12]);
    unsigned char *out = *static_cast<unsigned char **>(args[13]);
    REQUIRE(sgtd_stub_block_size(out) >= frame_pack_bytes(cn));
    u32 *w = reinterpret_cast<u32 *>(out);
    memcpy(w, *static_cast<const u32 **>(args[0]), 12 * sizeof(u32));      // (the batch's counters and flags, as the kernel copies them)
    if (S.frame_overflow > 0) { S.frame_overflow--; w[10] = 1; return; }
    if (S.frame_inliers >= 0) {
      w[12] = 1; w[13] = 100; w[14] = (u32)S.frame_inliers; w[15] = 64;
      long long *po = reinterpret_cast<long long *>(out + 72 + (size_t)cn * 8);
      double *sc = reinterpret_cast<double *>(po + cn + 1), *ps = sc + cn;
      long long *io = reinterpret_cast<long long *>(ps + (size_t)cn * 12);
      for (int k = 1; k <= cn; k++) { po[k] = S.frame_inliers; io[k] = S.frame_inliers; }
      sc[0] = (double)S.frame_inliers;
    }
  }
}

struct Descs {
  std::vector<double> side, angle, center;
  std::vector<float> vertex;
  std::vector<int32_t> label, node;
  std::vector<uint32_t> frame;
  sgtd_desc_soa soa() { return sgtd_desc_soa{side.data(), angle.data(), center.data(), vertex.data(), label.data(), frame.data(), node.data()}; }
  void resize(size_t n) { side.resize(n * 3); angle.resize(n * 3); center.resize(n * 3); vertex.resize(n * 9); label.resize(n * 3); node.resize(n * 3); frame.resize(n); }
};
Descs random_descs(std::mt19937 &rng, size_t n, uint32_t frame) {
  Descs d;
  d.resize(n);
  std::uniform_real_distribution<double> u(2.0, 28.0);
  for (size_t i = 0; i < n; i++) {
    double s[3] = {u(rng), u(rng), u(rng)};
    if (s[0] > s[1]) std::swap(s[0], s[1]);
    if (s[1] > s[2]) std::swap(s[1], s[2]);
    if (s[0] > s[1]) std::swap(s[0], s[1]);
    for (int k = 0; k < 3; k++) { d.side[i * 3 + k] = s[k]; d.label[i * 3 + k] = 3 + (int)(rng() % 9); d.node[i * 3 + k] = (int)(rng() % 200); d.center[i * 3 + k] = u(rng); }
    for (int k = 0; k < 9; k++) d.vertex[i * 9 + k] = (float)u(rng);
    d.frame[i] = frame;
  }
  return d;
}
}  // namespace

int main(int argc, char **argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 40;
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  std::mt19937 rng(20261003);
  Scenario S;
  sgtd_stub_set_launch_hook(hook, &S);
  sgtd_config cfg;
  sgtd_default_config(&cfg);
  S.cand_num = cfg.candidate_num;
  const int cn = cfg.candidate_num;

  // ---- a table from host descriptors, frame by frame; finalize; appends into a tail; more appends until the tail is merged
  sgtd_handle h = nullptr;
  OK(sgtd_create(&cfg, &h));
  g_last = h;
  for (int f = 0; f < frames; f++) {
    Descs d = random_descs(rng, 300 + rng() % 500, (uint32_t)f);
    sgtd_desc_soa s = d.soa();
    OK(sgtd_add(h, &s, (int64_t)d.frame.size()));
  }
  OK(sgtd_finalize(h));
  OK(sgtd_finalize(h));       // idempotent
  sgtd_stats st;
  OK(sgtd_get_stats(h, &st));
  REQUIRE(st.n_frames == frames && st.n_entries > 0);
  for (int f = frames; f < frames + 3; f++) {
    Descs d = random_descs(rng, 200, (uint32_t)f);
    sgtd_desc_soa s = d.soa();
    OK(sgtd_add(h, &s, 200));
    OK(sgtd_finalize(h));
  }
  OK(sgtd_get_stats(h, &st));
  REQUIRE(st.tail_entries > 0);

  // ---- batches of host descriptors: a clean one; one whose sweep outgrows the record buffer twice (re-runs, growth); one whose
  // reservations outran the buffer (the rate drops); one whose candidate pairs outgrow theirs (the list pass alone is re-run)
  Descs q = random_descs(rng, 700, (uint32_t)(frames + 3));
  sgtd_desc_soa qs = q.soa();
  std::vector<int32_t> n_cand(1), cf((size_t)cn), cv((size_t)cn);
  std::vector<int64_t> po((size_t)cn + 1);
  S.pairs_total = 5000;
  OK(sgtd_query_descs(h, &qs, 700));
  OK(sgtd_result_candidates(h, n_cand.data(), cf.data(), cv.data(), po.data()));
  S.sweep_overflows = 2; S.need = 3000000;
  OK(sgtd_query_descs(h, &qs, 700));
  OK(sgtd_sync(h));
  OK(sgtd_get_stats(h, &st));
  REQUIRE(st.overflowed == 1 && st.reruns_total >= 2 && S.sweep_overflows == 0);
  S.sweep_overflows = 1; S.need = 10; S.reservations = true;
  OK(sgtd_query_descs(h, &qs, 700));
  OK(sgtd_sync(h));
  S.reservations = false;
  S.pair_overflows = 1; S.pairs_total = 40000000;
  OK(sgtd_query_descs(h, &qs, 700));
  OK(sgtd_sync(h));
  OK(sgtd_get_stats(h, &st));
  REQUIRE(st.rewrites_total >= 1);
  // the verification and what follows it, on the batch's (stand-in) 4e7 pairs
  OK(sgtd_verify(h));
  std::vector<double> score((size_t)cn), pose((size_t)cn * 12);
  OK(sgtd_result_verify(h, 0, score.data(), pose.data()));
  int32_t bc = 0, bf = 0;
  double bs = 0;
  OK(sgtd_search_loop(h, 0.4, &bc, &bf, &bs));
  {
    std::vector<int64_t> off((size_t)cn + 1);
    std::vector<int32_t> qi(64);
    Descs ent; ent.resize(64);
    sgtd_desc_soa es = ent.soa();
    int64_t n_pairs = -1;
    OK(sgtd_result_inlier_entries(h, 0, off.data(), qi.data(), &es, 64, &n_pairs));
    REQUIRE(n_pairs == 0);
    std::vector<int32_t> pq(16); std::vector<int64_t> pe(16);
    int64_t np = 0;
    const int stp = sgtd_result_pairs(h, 0, pq.data(), pe.data(), 16, &np);
    REQUIRE(stp == SGTD_OK || stp == SGTD_ERR_CAPACITY);
  }
  // the hooks that shrink the work buffers (first batches overflow for real in the tests): honoured by a fresh handle
  S.pairs_total = 100;

  // ---- sgtd_search_frame: the one-wait path, too little room for the inlier pairs, more inlier pairs than the gather's first room
  // (the gather is enqueued again with room), a frame whose batch outgrew a work buffer (falls back to the separate calls)
  {
    std::vector<int32_t> fcf((size_t)cn), fcv((size_t)cn), fqi(40000);
    std::vector<int64_t> fpo((size_t)cn + 1), fio((size_t)cn + 1);
    std::vector<double> fsc((size_t)cn), fps((size_t)cn * 12);
    Descs ent; ent.resize(40000);
    sgtd_frame_search io{};
    io.cand_frame = fcf.data(); io.cand_votes = fcv.data(); io.pair_off = fpo.data(); io.score = fsc.data(); io.pose = fps.data();
    io.inlier_off = fio.data(); io.inlier_q_idx = fqi.data(); io.entries = ent.soa();
    S.frame_inliers = 0; io.capacity = 40000;
    OK(sgtd_search_frame(h, &qs, 700, &io));
    REQUIRE(io.n_inliers == 0 && io.n_cand == 1);
    S.frame_inliers = 100; io.capacity = 10;
    REQUIRE(sgtd_search_frame(h, &qs, 700, &io) == SGTD_ERR_CAPACITY && io.n_inliers == 100);
    io.capacity = 40000;
    OK(sgtd_search_frame(h, &qs, 700, &io));
    REQUIRE(io.n_inliers == 100);
    S.frame_inliers = 30000;                     // beyond the gather's first room of 16 384
    OK(sgtd_search_frame(h, &qs, 700, &io));
    REQUIRE(io.n_inliers == 30000);
    S.frame_inliers = 5; S.frame_overflow = 1;
    OK(sgtd_search_frame(h, &qs, 700, &io));
    S.frame_inliers = -1;
    OK(sgtd_get_stats(h, &st));
    REQUIRE(st.batches_total >= 0);
    // the caller's arrays page-locked (sgtd_host_alloc): the gather is handed them as they are — each must hold `capacity` entries.
    // One array shorter than that (the caller's mistake would be a kernel writing past it): the engine must see it and take its own block
    g_gathers_into.clear();
    void *pl[8];
    const size_t room = 2000, widths[8] = {24, 24, 24, 36, 12, 4, 12, 4};
    for (int k = 0; k < 8; k++) OK(sgtd_host_alloc(room * widths[k], &pl[k]));
    sgtd_frame_search d = io;
    d.entries = sgtd_desc_soa{(double *)pl[0], (double *)pl[1], (double *)pl[2], (float *)pl[3], (int32_t *)pl[4], (uint32_t *)pl[5], (int32_t *)pl[6]};
    d.inlier_q_idx = (int32_t *)pl[7];
    d.capacity = (int64_t)room; S.frame_inliers = 100;
    OK(sgtd_search_frame(h, &qs, 700, &d));
    REQUIRE(d.n_inliers == 100 && !g_gathers_into.empty() && g_gathers_into.back() == pl[0]);
    d.capacity = (int64_t)room + 1;             // one more than the arrays hold
    OK(sgtd_search_frame(h, &qs, 700, &d));
    REQUIRE(g_gathers_into.back() != pl[0]);
    d.capacity = (int64_t)room; d.entries.vertex = ent.vertex.data();       // one ordinary array among them
    OK(sgtd_search_frame(h, &qs, 700, &d));
    REQUIRE(g_gathers_into.back() != pl[0]);
    // candidate_selector alone in the one call (SGTD_FRAME_LISTS_ONLY): no verification is enqueued, the pairs are the lists' own; its
    // three ways out (one wait, too little room, the fall-back after an overflow) with page-locked and with ordinary arrays
    d.entries.vertex = (float *)pl[3];
    d.flags = SGTD_FRAME_LISTS_ONLY; d.capacity = (int64_t)room; S.frame_inliers = 100;
    OK(sgtd_search_frame(h, &qs, 700, &d));
    REQUIRE(d.n_inliers == 100 && g_gathers_into.back() == pl[0]);
    d.capacity = 10;
    REQUIRE(sgtd_search_frame(h, &qs, 700, &d) == SGTD_ERR_CAPACITY && d.n_inliers == 100);
    d.capacity = (int64_t)room; S.frame_overflow = 1;
    OK(sgtd_search_frame(h, &qs, 700, &d));
    io.flags = SGTD_FRAME_LISTS_ONLY; io.capacity = 40000; S.frame_inliers = 30000;
    OK(sgtd_search_frame(h, &qs, 700, &io));
    REQUIRE(io.n_inliers == 30000);
    io.flags = 0;
    S.frame_inliers = -1;
    for (int k = 0; k < 8; k++) OK(sgtd_host_free(pl[k]));
  }

  // ---- views: a second handle on the same table; its pending batch after the owner's table changed; destroy order
  sgtd_handle v = nullptr;
  OK(sgtd_create(&cfg, &v));
  OK(sgtd_attach_table(v, h));
  OK(sgtd_query_descs(v, &qs, 700));
  OK(sgtd_sync(v));
  OK(sgtd_query_descs(v, &qs, 700));          // pending ...
  {
    Descs d = random_descs(rng, 150, (uint32_t)(frames + 4));
    sgtd_desc_soa s = d.soa();
    OK(sgtd_add(h, &s, 150));                 // ... while the owner's table changes (its buffers are reallocated)
    OK(sgtd_finalize(h));
  }
  REQUIRE(sgtd_verify(v) == SGTD_ERR_STATE);
  REQUIRE(sgtd_query_descs(v, &qs, 700) == SGTD_ERR_STATE);
  { int64_t e0[4] = {0, 1, 2, 3}; Descs o4; o4.resize(4); sgtd_desc_soa os = o4.soa(); REQUIRE(sgtd_fetch_entries(v, e0, 4, &os) == SGTD_ERR_STATE); }
  REQUIRE(sgtd_destroy(h) == SGTD_ERR_STATE);  // a view is still attached
  OK(sgtd_attach_table(v, h));
  OK(sgtd_query_descs(v, &qs, 700));
  OK(sgtd_verify(v));
  { Descs d = random_descs(rng, 10, 0); sgtd_desc_soa s = d.soa(); REQUIRE(sgtd_add(v, &s, 10) == SGTD_ERR_STATE); }
  // the owner's own batches on a tail: the fifth merges the tail (a rebuild) — the view must notice
  for (int b = 0; b < 6; b++) { OK(sgtd_query_descs(h, &qs, 700)); OK(sgtd_sync(h)); }
  OK(sgtd_get_stats(h, &st));
  REQUIRE(st.tail_entries == 0);
  REQUIRE(sgtd_query_descs(v, &qs, 700) == SGTD_ERR_STATE);
  OK(sgtd_destroy(v));

  // ---- the table file: save, load into a fresh handle (header checks, sizes), append, query; damaged files refuse
  const std::string path = dir + "/engine_driver_table.bin";
  OK(sgtd_save_table(h, path.c_str()));
  sgtd_handle l = nullptr;
  OK(sgtd_create(&cfg, &l));
  OK(sgtd_load_table(l, path.c_str()));
  sgtd_stats sl;
  OK(sgtd_get_stats(l, &sl));
  OK(sgtd_get_stats(h, &st));
  REQUIRE(sl.n_entries == st.n_entries && sl.n_frames == st.n_frames);
  { Descs d = random_descs(rng, 120, (uint32_t)(frames + 5)); sgtd_desc_soa s = d.soa(); OK(sgtd_add(l, &s, 120)); }
  OK(sgtd_query_descs(l, &qs, 700));
  OK(sgtd_sync(l));
  {
    FILE *f = fopen(path.c_str(), "rb");
    REQUIRE(f);
    std::vector<unsigned char> bytes;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + got);
    fclose(f);
    const std::string cut = dir + "/engine_driver_table_cut.bin";
    for (size_t keep : {(size_t)0, (size_t)7, (size_t)60, bytes.size() / 2, bytes.size() - 1}) {
      f = fopen(cut.c_str(), "wb"); REQUIRE(f);
      fwrite(bytes.data(), 1, keep, f); fclose(f);
      sgtd_handle t = nullptr;
      OK(sgtd_create(&cfg, &t));
      REQUIRE(sgtd_load_table(t, cut.c_str()) != SGTD_OK);
      OK(sgtd_destroy(t));
    }
    remove(cut.c_str());
  }
  OK(sgtd_destroy(l));
  remove(path.c_str());

  // ---- two "devices" behind one handle (both shards on the stand-in's device 0): add, finalize, a batch, the verification
  {
    const int ids[2] = {0, 0};
    sgtd_handle m = nullptr;
    OK(sgtd_create_multi(&cfg, ids, 2, &m));
    REQUIRE(sgtd_device_count(m) == 2);
    for (int f = 0; f < 12; f++) { Descs d = random_descs(rng, 250, (uint32_t)f); sgtd_desc_soa s = d.soa(); OK(sgtd_add(m, &s, 250)); }
    OK(sgtd_finalize(m));
    OK(sgtd_query_descs(m, &qs, 700));
    OK(sgtd_result_candidates(m, n_cand.data(), cf.data(), cv.data(), po.data()));
    OK(sgtd_verify(m));
    OK(sgtd_result_verify(m, 0, score.data(), pose.data()));
    OK(sgtd_destroy(m));
  }

  OK(sgtd_destroy(h));
  REQUIRE(sgtd_stub_device_blocks() == 0 && sgtd_stub_device_bytes() == 0);       // every device buffer was freed
  printf("engine host code under the sanitizers: ok (%llu launches, %llu sweeps, %llu frame packs, device peak %.1f MB)\n",
         sgtd_stub_launches(), S.sweeps, S.packs, sgtd_stub_device_peak() / 1048576.0);
  return 0;
}
