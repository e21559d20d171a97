// sgtd_accel.hip — C ABI (include/sgtd_accel.h) over the gfx950 kernels.
// Host side: buffer ownership, launch order, result marshalling.  No CPU
// compute path exists here: every entry point needs the HIP device.
#include "../../include/sgtd_accel.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "build_kernel.hip.h"
#include "common.hip.h"
#include "table_kernels.hip.h"
#include "probe_kernels.hip.h"
#include "select_kernels.hip.h"
#include "verify_kernels.hip.h"
#include "verify_mfma.hip.h"
#include "exchange_kernels.hip.h"
#include "graph_ingest.hip.h"
#include "table_file.h"

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  bool borrowed = false;     // the memory belongs to another handle (sgtd_attach_table): never freed or regrown here
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct DescStore {
  DevBuf side, angle, center, vertex, label, frame, node_id, qrec;
  size_t cap = 0;
  bool with_thr2 = false;   // query descriptors carry their squared match threshold
  DescArrays view() const {
    DescArrays a;
    a.side = side.as<double>(); a.angle = angle.as<double>(); a.center = center.as<double>();
    a.vertex = vertex.as<float>(); a.label = label.as<int>(); a.frame = frame.as<u32>();
    a.node_id = node_id.as<int>();
    a.qrec = with_thr2 ? qrec.as<QueryRec>() : nullptr;
    return a;
  }
};

// A host array in page-locked memory (grow-only): the per-batch result tables land here by direct DMA — a copy into
// pageable memory goes through the runtime's staging at about a third of the rate (and ~150 us each on this runtime).
template <class T>
struct PinnedVec {
  T *p = nullptr;
  size_t n = 0, cap = 0;
  PinnedVec() = default;
  PinnedVec(const PinnedVec &) = delete;
  PinnedVec &operator=(const PinnedVec &) = delete;
  ~PinnedVec() { if (p) (void)hipHostFree(p); }
  bool resize(size_t want) {
    if (want > cap) {
      void *np = nullptr;
      const size_t c = std::max<size_t>(want + want / 4, 64);
      if (hipHostMalloc(&np, c * sizeof(T), hipHostMallocDefault) != hipSuccess) return false;
      if (p) (void)hipHostFree(p);      // (contents are not kept: every user refills after a resize)
      p = static_cast<T *>(np);
      cap = c;
    }
    n = want;
    return true;
  }
  T *data() { return p; }
  const T *data() const { return p; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
};

constexpr size_t kCtrWords = 1024 + 8 * 1024;   // ProbeBuffers::ctr: counters + the sweep's ticket-queue heads
enum { EV_START = 0, EV_BUILD, EV_SORT, EV_PROBE, EV_VOTES, EV_TOPK, EV_COUNT_T, EV_SCAN, EV_WRITE, EV_COUNT };

}  // namespace

namespace multi { struct Group; }

struct sgtd_engine {
  multi::Group *grp = nullptr;   // set: this handle is a group of per-device engines (multi.hip.h)
  sgtd_config cfg;
  DevCfg dc;
  hipStream_t stream = nullptr;
  std::string err;
  int n_cus = 256;
  bool timing = false;
  hipEvent_t ev[EV_COUNT] = {};

  // ---- a handle may borrow the finalized table of another handle on the same device (sgtd_attach_table): its own
  // work buffers, results and stream, the owner's table — two batches in flight over one map
  sgtd_engine *attached_to = nullptr;
  unsigned long long table_version = 0;   // bumped by everything that changes the table or its probe layout
  unsigned long long attached_version = 0;
  int n_views = 0;                        // handles attached to this one's table
  // ---- table, insertion order (cold)
  u32 current_frame_id = 0;
  DescStore tab;
  int64_t n_entries = 0;
  int64_t n_add_calls = 0;
  bool have_frames = false;
  u32 frame_lo = 0, frame_hi = 0;
  bool finalized = false;  // the first query builds the (possibly empty) bucket directory

  // ---- table, probe layout (hot): a main segment and, after appends to a finalized table, a
  // tail segment — each a complete probe layout (HotEntry[n] 16 B, perm[n] for the table dump, bucket directory,
  // key hash) over a contiguous range of insertion indices.  Appending re-sorts only the tail;
  // the tail is merged into the main segment when it outgrows an eighth of it.
  struct Segment {
    DevBuf hot, perm, hash, bucket_start, bucket_key, dir;
    u32 hash_mask = 0;
    int64_t n_buckets = 0;
    int64_t g0 = 0, g1 = 0;              // insertion indices [g0, g1)
    double sum_len_sq = 0.0;             // sum over buckets of len^2 (sizes the first batch's work buffers)
    u32 frame_hi = 0;                    // largest frame id inside (at build time)
    bool built = false;
  };
  Segment seg[2];
  int n_seg = 1;
  u32 append_min_frame = 0xFFFFFFFFu;    // smallest frame id appended since the main segment was built
  int64_t tail_max = 0;                  // SGTD_TAIL_MAX: entries the tail may hold before a merge (0 = an eighth of the main segment)
  float ms_finalize = 0.f;               // wall time of the last probe-layout build
  u32 coarse_at = 62, whole_at = 62;     // SGTD_COARSE_AT, SGTD_WHOLE_AT: see TableView
  long long thr2_pending = 0;            // descriptors handed in by the caller whose QueryRec is still to be written (launch_select)
  u32 rec_rate_hook = 0;                 // SGTD_REC_RATE (test hook): ProbeBuffers::rec_rate, 1..256
  u32 rec_rate_cap = 256;                // upper bound on rec_rate for the pending batch: quartered by every re-run whose RESERVATIONS
                                         // (not its matches) outgrew the record buffer — long visit lists with few matches, skewed maps
  bool wide_pairs = false;               // SGTD_WIDE_PAIRS (test hook): 8-byte compact words whatever the ids' rank bits
  int tail_batches = 0;                  // query batches swept with the current tail (it is merged after a few: see settle_tail)
  DevBuf slice_of, sq_sum;
  // entry ids of the probe layout (common.hip.h IdMap), rebuilt over the whole table by every finalize
  DevBuf frame_first, by_frame, id_of_g, longest;
  u32 id_bits = 0;                       // bits of the in-frame rank the built segments' ids use (0: none built)
  bool id_by_frame = false;              // frames out of insertion order: by_frame / id_of_g are in use
  // sort scratch
  DevBuf keyA, keyB, valA, valB, hist, digit_tot, flags, bad_flag;
  std::vector<DevBuf> scan_lvl;

  // ---- build scratch
  DevBuf kp_off_dev, xyz_dev, label_dev, ws_keys, ws_slots, cnt_scan;
  DevBuf b_kp_off_dev, b_xyz_dev, b_label_dev;   // sgtd_build's own staging (never the pending batch's inputs)
  DescStore tmp;        // strided build output for map construction
  DescStore fetch;      // sgtd_fetch_entries staging
  DevBuf fetch_idx;
  DevBuf tmp_count;

  // ---- query batch
  DescStore qd;         // strided query descriptors
  DevBuf q_count;
  long long q_stride = 0;
  int nq = 0;
  bool batch_valid = false, batch_synced = false;
  // inputs of the last batch (for a re-run after a work-buffer overflow)
  int last_kind = 0;    // 1 frames, 2 descs
  const float *last_xyz = nullptr;
  const u32 *last_label = nullptr;
  std::vector<long long> last_kp_off;
  int last_max_n = 0;
  u32 last_qframe = 0;  // current_frame_id_ when the batch was enqueued (a re-run stamps the same id)
  DevBuf totals;        // 3 x u64, zeroed once: batch_totals_kernel's running counts
  DevBuf cursors, list, n_visit, votes, slot_of;   // cursors: the batch's counters, overflow flags and ticket heads (ProbeBuffers::ctr)
  DevBuf q_M, q_P, q_pairs, q_pair_base, blk_count, rec, rec_cell, rec_dis;
  DevBuf c_pair, c_blk;           // compact candidate-match lists between block_count and block_write
  DevBuf amb_queue;               // provisional records awaiting the exact test
  DevBuf n_cand, cand_frame, cand_votes, pair_off, pairs;
  DevBuf v_score, v_pose, v_inlier, v_best;   // sgtd_verify results of the batch
  DevBuf inl_pairs, inl_off;                  // sgtd_result_inlier_pairs staging
  bool verify_counted = false;                // the last verify_enqueue left the inlier counts in inl_counts
  DevBuf inl_counts;                          // sgtd_search_frame: inlier pairs per candidate
  DevBuf in_block;                            // copy_in: a frame's descriptors as they arrive, one block
  size_t frame_inl_cap = 0;                   // ... entries its gather has room for (grown when a frame has more inlier pairs)
  char *frame_host = nullptr;                 // ... page-locked host memory the call's last kernels write straight into: the packed small results,
  size_t frame_host_bytes = 0;                //     then the inlier pairs' entries field by field and their query indices (no copy, no second wait)
  DevBuf b_in, b_out;                         // sgtd_build's one-block staging on the device: inputs / all descriptor fields
  char *pin_build = nullptr;                  // ... and in page-locked host memory
  size_t pin_build_cap = 0;
  DevBuf v_hyp64, v_hyp32, v_bound;           // hypotheses between the two passes of sgtd_verify
  DevBuf v_hypB, v_tau, v_words;              // the matrix-core vote pass: hypothesis features, |t|^2, vote words
  DevBuf v_okey[2], v_oval[2];                // its dispatch order: (candidate frame, candidate index) sorted
  bool verified = false;
  // ---- multi-GPU step (exchange_kernels.hip.h): the batch's local candidate tables are written, packed, into a caller
  // device buffer as soon as they are final (behind votes_topk_kernel / topk_kernel) and ev_cand is recorded; a side
  // stream waits for it (sgtd_export_wait), all-gathers and merges while the match lists are written, and records
  // ev_export_free (sgtd_export_release): the next batch's export waits for that before it overwrites the buffer
  int *export_packed = nullptr;
  size_t export_cap = 0;                      // ints the caller's buffer holds
  hipEvent_t ev_cand = nullptr, ev_export_free = nullptr;
  bool export_busy = false;
  u32 batch_serial = 0;
  bool defer_lists = false;                   // sgtd_set_deferred_lists: a batch stops behind the candidate tables, sgtd_finish_lists writes the lists
  bool lists_pending = false;                 // ... and has not been called for the batch yet
  bool lists_masked = false;                  // the batch's lists were written for a subset of its candidates (pair_off holds the masked offsets)
  bool fused_votes_last = false;              // the batch's candidate tables came from votes_topk_kernel (it also wrote the unmasked offsets)
  const u64 *verify_keep = nullptr;           // sgtd_verify_masked: device mask of the candidates to verify (one launch)
  const u64 *list_keep = nullptr;             // the mask the batch's lists were last written with (a re-run of the list pass: the same)
  bool pairs_per_query = false;               // the batch's match lists were written by pairs_query_kernel (a re-run of the write pass: the same)
  DevBuf rough_qi, rough_entry, rough_frame, rough_cell, rough_dis;
  size_t rec_cap = (size_t)1 << 25;    // match records (grown on overflow)
  bool rec_cap_fixed = false;          // SGTD_REC_CAP given: no pre-sizing from the table statistics
  size_t pair_cap = (size_t)1 << 24;   // candidate pairs (grown on overflow)
  DevBuf n_valid, cell_rows, gid, q_prefix, group_first, n_groups;
  DevBuf pos_of_slot, rec_off, pass_pool;   // pass slots and records (probe_kernels.hip.h)
  size_t pool_units = 0;                    // pass pool capacity in 16-B units (0: sized by the first batch; grown on overflow)
  size_t group_cap = 0;                     // GroupRows reserved (0: sized by the first batch; grown on overflow)
  size_t group_cap_hook = 0;                // SGTD_GROUP_CAP (test hook): the first reservation
  int sorted_chunk = 0;                // > 0: fixed descriptors per ticket (SGTD_SORTED_CHUNK), else adaptive
  bool diag = false;                   // diagnostic probe build: cell index + distance per match
  int select_mode = 0;                 // SGTD_SELECT_MODE (test hook): 0 auto, 1 the five-kernel passes over the records, 2 the per-query
                                       // workgroups (select_kernels.hip.h) wherever they are supported, whatever the batch size
  // host copies after sync
  PinnedVec<u32> h_count, h_pair_base, h_q_M;
  PinnedVec<unsigned long long> h_q_P;
  PinnedVec<int> h_n_cand, h_cand_frame, h_cand_votes;
  PinnedVec<long long> h_pair_off;
  sgtd_stats stats{};
  // pinned host staging of the small transfers (see d2h / h2d below)
  char *pin = nullptr;
  size_t pin_cap = 0, pin_used = 0;
  struct PinCopy { void *dst; size_t off, bytes; };
  std::vector<PinCopy> pin_pending;
};

#include "multi.hip.h"

namespace {

#define HIPCHK(call)                                                          \
  do {                                                                        \
    hipError_t _s = (call);                                                   \
    if (_s != hipSuccess) {                                                   \
      e->err = std::string(#call) + ": " + hipGetErrorString(_s);             \
      return SGTD_ERR_HIP;                                                    \
    }                                                                         \
  } while (0)
#define CHK(call)                 \
  do {                            \
    int _r = (call);              \
    if (_r != SGTD_OK) return _r; \
  } while (0)

int ensure(sgtd_engine *e, DevBuf &b, size_t bytes, bool keep = false) {
  if (bytes <= b.bytes) return SGTD_OK;
  if (b.borrowed) { e->err = "a table attached from another handle cannot grow here"; return SGTD_ERR_STATE; }
  // (a buffer that grows again gets a quarter more than asked for: the one-frame-per-call pattern would
  // otherwise free and allocate a dozen work buffers on every frame that is a little larger than the last)
  size_t want = keep ? std::max(bytes, b.bytes + b.bytes / 2) : (b.p ? bytes + bytes / 4 : bytes);
  // nothing to keep: the old buffer goes first (a record buffer that grows from 30 to 45 GB must not need 75 for a moment)
  if (!keep && b.p) {
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
  }
  void *np = nullptr;
  {
    const hipError_t st_ = hipMalloc(&np, want);
    if (st_ != hipSuccess) {
      (void)hipGetLastError();      // (the runtime's "last error" is the caller's too — torch raises on it at its next call — this one is reported here)
      e->err = "hipMalloc of " + std::to_string(want) + " bytes: " + hipGetErrorString(st_);
      return SGTD_ERR_HIP;
    }
  }
  e->stats.device_allocs_total++;
  if (keep && b.p && b.bytes) {
    HIPCHK(hipMemcpyAsync(np, b.p, b.bytes, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  if (b.p) HIPCHK(hipFree(b.p));
  b.p = np;
  b.bytes = want;
  return SGTD_OK;
}

int ensure_store(sgtd_engine *e, DescStore &s, size_t cap, bool keep = false) {
  if (cap <= s.cap) return SGTD_OK;
  size_t want = keep ? std::max(cap, s.cap + s.cap / 2) : cap;
  CHK(ensure(e, s.side, want * 3 * sizeof(double), keep));
  CHK(ensure(e, s.angle, want * 3 * sizeof(double), keep));
  CHK(ensure(e, s.center, want * 3 * sizeof(double), keep));
  CHK(ensure(e, s.vertex, want * 9 * sizeof(float), keep));
  CHK(ensure(e, s.label, want * 3 * sizeof(int), keep));
  CHK(ensure(e, s.frame, want * sizeof(u32), keep));
  CHK(ensure(e, s.node_id, want * 3 * sizeof(int), keep));
  if (s.with_thr2) CHK(ensure(e, s.qrec, want * sizeof(QueryRec), keep));
  s.cap = want;
  return SGTD_OK;
}

void free_buf(DevBuf &b) {
  if (b.p && !b.borrowed) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  b.borrowed = false;
}
void free_store(DescStore &s) {
  free_buf(s.side); free_buf(s.angle); free_buf(s.center); free_buf(s.vertex);
  free_buf(s.label); free_buf(s.frame); free_buf(s.node_id); free_buf(s.qrec);
  s.cap = 0;
}

// ---------------------------------------------------------------------------
// Small host <-> device transfers go through ONE pinned staging buffer per handle: a
// hipMemcpyAsync on pageable memory costs ~150 us each on this runtime (the reference's
// one-frame-per-call pattern issues ~60 of them per frame), on pinned memory a few.
// d2h: queue a copy device -> pinned and remember where the caller wants it; xfer_sync: wait for
// the stream, then move the queued results to the caller's (pageable) buffers.  h2d: copy the
// caller's data into pinned memory at once, queue pinned -> device; the pinned bytes are reused
// only after the next xfer_sync.  Transfers beyond kPinMax take the direct path (bandwidth-bound).
// ---------------------------------------------------------------------------
constexpr size_t kPinMax = (size_t)1 << 20, kPinCap = (size_t)16 << 20;   // (a frame's 7 200 descriptors: 172 KB of sides, 259 KB of vertices — one frame's fields all take
                                                                           // the staged path; beyond 1 MB a transfer is bandwidth-bound and the extra host copy costs more than it saves)

int xfer_sync(sgtd_engine *e) {
  HIPCHK(hipStreamSynchronize(e->stream));
  for (const auto &c : e->pin_pending) std::memcpy(c.dst, e->pin + c.off, c.bytes);
  e->pin_pending.clear();
  e->pin_used = 0;
  return SGTD_OK;
}

// d2h queues {destination, pinned offset}; only xfer_sync applies the queue.  A function that returns early (a failed
// HIP call between its d2h calls and its xfer_sync) must not leave entries behind whose destinations are its own
// locals: PinScope drops whatever is still queued when the function is left.
struct PinScope {
  sgtd_engine *e;
  explicit PinScope(sgtd_engine *e_) : e(e_) {}
  ~PinScope() {
    if (e && !e->pin_pending.empty()) {
      (void)hipStreamSynchronize(e->stream);      // (the copies into the pinned buffer may still be in flight)
      e->pin_pending.clear();
      e->pin_used = 0;
    }
  }
};

int pin_room(sgtd_engine *e, size_t bytes, size_t *off) {
  if (!e->pin) {
    void *p = nullptr;
    HIPCHK(hipHostMalloc(&p, kPinCap, hipHostMallocDefault));
    e->pin = static_cast<char *>(p);
    e->pin_cap = kPinCap;
  }
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (e->pin_used + need > e->pin_cap) CHK(xfer_sync(e));   // full: finish what is queued, start over
  *off = e->pin_used;
  e->pin_used += need;
  return SGTD_OK;
}

int d2h(sgtd_engine *e, void *dst, const void *src_dev, size_t bytes) {
  if (bytes == 0) return SGTD_OK;
  if (bytes > kPinMax) {
    HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, e->stream));
    return SGTD_OK;
  }
  size_t off;
  CHK(pin_room(e, bytes, &off));
  HIPCHK(hipMemcpyAsync(e->pin + off, src_dev, bytes, hipMemcpyDeviceToHost, e->stream));
  e->pin_pending.push_back({dst, off, bytes});
  return SGTD_OK;
}

int h2d(sgtd_engine *e, void *dst_dev, const void *src, size_t bytes) {
  if (bytes == 0) return SGTD_OK;
  if (bytes > kPinMax) {
    HIPCHK(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, e->stream));
    return SGTD_OK;
  }
  size_t off;
  CHK(pin_room(e, bytes, &off));
  std::memcpy(e->pin + off, src, bytes);
  HIPCHK(hipMemcpyAsync(dst_dev, e->pin + off, bytes, hipMemcpyHostToDevice, e->stream));
  return SGTD_OK;
}

int grid_for(long long n, int threads) { return (int)((n + threads - 1) / threads); }
// what 32-bit indices can name: candidate pairs one by one, match records by granules of four (probe_kernels.hip.h)
constexpr size_t kIndexLimit = 0xFFFFFFF0ull;
constexpr size_t kRecLimit = kIndexLimit << SGTD_REC_SHIFT;
// dynamic LDS a kernel may ask for: gfx950 gives a workgroup 160 KB, the kernel's static __shared__ arrays included (the
// policy bound of 150 KB alone let a launch of a kernel with 12.6 KB of static LDS fail for spans that need 147.5 .. 150 KB)
template <class K>
size_t lds_room(K *kernel) {
  hipFuncAttributes a;
  const size_t fixed = hipFuncGetAttributes(&a, reinterpret_cast<const void *>(kernel)) == hipSuccess ? a.sharedSizeBytes : 16384;
  return std::min<size_t>(150 * 1024, 160 * 1024 - std::min<size_t>(fixed, 160 * 1024));
}

// device-wide exclusive scan (u32), in place allowed
int device_scan(sgtd_engine *e, const u32 *in, u32 *out, long long n, size_t lvl = 0) {
  if (n <= 0) return SGTD_OK;
  const long long per = (long long)SGTD_SCAN_THREADS * SGTD_SCAN_ITEMS;
  const long long nb = (n + per - 1) / per;
  if (nb == 1) {
    scan_apply_kernel<<<1, SGTD_SCAN_THREADS, 0, e->stream>>>(in, out, nullptr, n);
    HIPCHK(hipGetLastError());
    return SGTD_OK;
  }
  if (e->scan_lvl.size() <= lvl) e->scan_lvl.resize(lvl + 1);
  CHK(ensure(e, e->scan_lvl[lvl], (size_t)nb * sizeof(u32)));
  u32 *sums = e->scan_lvl[lvl].as<u32>();
  scan_reduce_kernel<<<(int)nb, SGTD_SCAN_THREADS, 0, e->stream>>>(in, sums, n);
  HIPCHK(hipGetLastError());
  CHK(device_scan(e, sums, sums, nb, lvl + 1));
  sums = e->scan_lvl[lvl].as<u32>();
  scan_apply_kernel<<<(int)nb, SGTD_SCAN_THREADS, 0, e->stream>>>(in, out, sums, n);
  HIPCHK(hipGetLastError());
  return SGTD_OK;
}

// ---------------------------------------------------------------------------
// BuildSingleScanSTD launch for a batch of frames (inputs already on device)
// ---------------------------------------------------------------------------
int launch_build(sgtd_engine *e, const float *d_xyz, const u32 *d_label, const long long *d_kp_off,
                 int n_frames, int max_n, u32 frame_id0, int frame_step, DescArrays out,
                 long long out_stride, u32 *out_count) {
  if (n_frames <= 0) return SGTD_OK;
  if (max_n > 65535) return SGTD_ERR_UNSUPPORTED;
  const int K = e->dc.K, tpi = e->dc.tpi;
  long long max_t = (long long)std::max(max_n, 1) * tpi;
  int max_slots = 64;
  while (max_slots < max_t + 1) max_slots <<= 1;
  const size_t lds_limit = 160 * 1024;
  size_t lds_full = build_lds_bytes(std::max(max_n, 1), K, tpi, true, max_slots);
  size_t lds_base = build_lds_bytes(std::max(max_n, 1), K, tpi, false, max_slots);
  BuildParams P;
  P.xyz = d_xyz; P.label = d_label; P.kp_off = d_kp_off; P.n_frames = n_frames;
  P.frame_id0 = frame_id0; P.frame_id_step = frame_step; P.out_stride = out_stride;
  P.out_count = out_count; P.max_n = std::max(max_n, 1); P.max_slots = max_slots;
  P.ws_keys = nullptr; P.ws_slots = nullptr;
  if (lds_full <= lds_limit) {
    int per_cu = (int)std::max<size_t>(1, lds_limit / lds_full);
    int grid = std::min(n_frames, e->n_cus * std::min(per_cu, 4));
#define SGTD_LAUNCH_BUILD(DEDUP, KM_, LDS_)                                                                    \
  do {                                                                                                          \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&build_frames_kernel<DEDUP, KM_>),                \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)));                       \
    build_frames_kernel<DEDUP, KM_><<<grid, SGTD_BUILD_THREADS, (LDS_), e->stream>>>(P, e->dc, out);            \
  } while (0)
#define SGTD_LAUNCH_BUILD_K(DEDUP, LDS_)                                                                        \
  do {                                                                                                          \
    if (K <= 4) SGTD_LAUNCH_BUILD(DEDUP, 4, LDS_);                                                              \
    else if (K <= 8) SGTD_LAUNCH_BUILD(DEDUP, 8, LDS_);                                                         \
    else if (K <= 10) SGTD_LAUNCH_BUILD(DEDUP, 10, LDS_);                                                       \
    else if (K <= 12) SGTD_LAUNCH_BUILD(DEDUP, 12, LDS_);                                                       \
    else SGTD_LAUNCH_BUILD(DEDUP, 16, LDS_);                                                                    \
  } while (0)
    SGTD_LAUNCH_BUILD_K(true, lds_full);
  } else if (lds_base <= lds_limit) {
    int grid = std::min(n_frames, e->n_cus * 2);
    CHK(ensure(e, e->ws_keys, (size_t)grid * max_t * sizeof(u64)));
    CHK(ensure(e, e->ws_slots, (size_t)grid * max_slots * sizeof(u32)));
    P.ws_keys = e->ws_keys.as<u64>();
    P.ws_slots = e->ws_slots.as<u32>();
    SGTD_LAUNCH_BUILD_K(false, lds_base);
#undef SGTD_LAUNCH_BUILD_K
#undef SGTD_LAUNCH_BUILD
  } else {
    return SGTD_ERR_UNSUPPORTED;
  }
  HIPCHK(hipGetLastError());
  return SGTD_OK;
}

// uploads keypoints (or takes device pointers) and the offsets; returns views
// own_set: sgtd_build stages into buffers of its own — a synced query batch may still be re-run
// from the batch's staging buffers (sgtd_result_rough's diagnostic pass)
int stage_inputs(sgtd_engine *e, const float *xyz, const u32 *label, const int64_t *kp_off,
                 int n_frames, int device_ptrs, const float **d_xyz, const u32 **d_label,
                 int *max_n, bool own_set = false) {
  DevBuf &kp_buf = own_set ? e->b_kp_off_dev : e->kp_off_dev;
  DevBuf &xyz_buf = own_set ? e->b_xyz_dev : e->xyz_dev;
  DevBuf &label_buf = own_set ? e->b_label_dev : e->label_dev;
  std::vector<long long> off(n_frames + 1);
  int mx = 0;
  for (int f = 0; f <= n_frames; f++) off[f] = kp_off[f];
  for (int f = 0; f < n_frames; f++) {
    long long n = off[f + 1] - off[f];
    if (n < 0 || n > 65535) return SGTD_ERR_INVALID;
    mx = std::max(mx, (int)n);
  }
  *max_n = mx;
  CHK(ensure(e, kp_buf, (size_t)(n_frames + 1) * sizeof(long long)));
  // through the page-locked staging (the bytes leave `off` here): no wait for the stream — a caller that enqueues batch
  // after batch keeps the device fed, the next batch's kernels queue up behind the running one's
  if ((size_t)(n_frames + 1) * sizeof(long long) <= kPinMax) {
    CHK(h2d(e, kp_buf.p, off.data(), (size_t)(n_frames + 1) * sizeof(long long)));
  } else {
    HIPCHK(hipMemcpyAsync(kp_buf.p, off.data(), (size_t)(n_frames + 1) * sizeof(long long), hipMemcpyHostToDevice, e->stream));
    CHK(xfer_sync(e));      // (the staging vector dies at return)
  }
  const long long total = off[n_frames] - off[0];
  if (device_ptrs) {
    *d_xyz = xyz;
    *d_label = label;
  } else {
    CHK(ensure(e, xyz_buf, (size_t)std::max<long long>(total + off[0], 1) * 3 * sizeof(float)));
    CHK(ensure(e, label_buf, (size_t)std::max<long long>(total + off[0], 1) * sizeof(u32)));
    if (total > 0) {
      CHK(h2d(e, xyz_buf.as<float>() + off[0] * 3, xyz + off[0] * 3,
                            (size_t)total * 3 * sizeof(float)));
      CHK(h2d(e, label_buf.as<u32>() + off[0], label + off[0], (size_t)total * sizeof(u32)));
    }
    *d_xyz = xyz_buf.as<float>();
    *d_label = label_buf.as<u32>();
  }
  return SGTD_OK;
}

// device SoA range -> caller SoA (NULL members skipped)
int copy_out(sgtd_engine *e, const DescStore &s, size_t first, size_t n, sgtd_desc_soa *out, size_t dst0) {
  if (n == 0) return SGTD_OK;
#define CP(field, T, w)                                                                       \
  if (out->field)                                                                             \
    CHK(d2h(e, out->field + dst0 * (w), s.field.as<T>() + first * (w),             \
                          n * (w) * sizeof(T)));
  CP(side, double, 3) CP(angle, double, 3) CP(center, double, 3) CP(vertex, float, 9)
  CP(label, int, 3) CP(frame, u32, 1) CP(node_id, int, 3)
#undef CP
  CHK(xfer_sync(e));
  return SGTD_OK;
}

int copy_in(sgtd_engine *e, DescStore &s, size_t first, size_t n, const sgtd_desc_soa *in, bool wait = true) {
  if (n == 0) return SGTD_OK;
  // a frame's descriptors (up to a few MB): the seven fields side by side in the page-locked staging, ONE transfer into a block on
  // the device and a kernel that deals the words to the arrays — seven transfers of ~100 KB took ~0.2 ms before the first kernel
  // of a one-frame call could start, one takes 0.04
  const DescBlockOffsets bo = desc_block_offsets(n);
  static const bool block_on = [] { const char *o = getenv("SGTD_COPY_IN_BLOCK"); return !(o && !atoi(o)); }();
  if (block_on && bo.total <= (4u << 20)) {
    size_t off;
    CHK(pin_room(e, bo.total, &off));
    const void *src[7] = {in->side, in->angle, in->center, in->vertex, in->label, in->frame, in->node_id};
    void *dst[7] = {s.side.as<double>() + first * 3, s.angle.as<double>() + first * 3, s.center.as<double>() + first * 3, s.vertex.as<float>() + first * 9,
                    s.label.as<int>() + first * 3, s.frame.as<u32>() + first, s.node_id.as<int>() + first * 3};
    const size_t bytes[7] = {n * 24, n * 24, n * 24, n * 36, n * 12, n * 4, n * 12};
    u32 present = 0;
    for (int f = 0; f < 7; f++) {
      if (src[f]) { std::memcpy(e->pin + off + bo.off[f], src[f], bytes[f]); present |= 1u << f; }
      else HIPCHK(hipMemsetAsync(dst[f], 0, bytes[f], e->stream));
    }
    CHK(ensure(e, e->in_block, bo.total));
    HIPCHK(hipMemcpyAsync(e->in_block.p, e->pin + off, bo.total, hipMemcpyHostToDevice, e->stream));
    unpack_desc_block_kernel<<<(unsigned)std::min<long long>(grid_for((long long)n * 34, 256), 512), 256, 0, e->stream>>>(e->in_block.as<unsigned char>(), (u32)n, present, s.view(),
                                                                                                                       (long long)first);
    HIPCHK(hipGetLastError());
    if (wait) CHK(xfer_sync(e));
    return SGTD_OK;
  }
#define CP(field, T, w)                                                                        \
  if (in->field)                                                                               \
    CHK(h2d(e, s.field.as<T>() + first * (w), in->field, n * (w) * sizeof(T)));               \
  else                                                                                         \
    HIPCHK(hipMemsetAsync(s.field.as<T>() + first * (w), 0, n * (w) * sizeof(T), e->stream));
  CP(side, double, 3) CP(angle, double, 3) CP(center, double, 3) CP(vertex, float, 9)
  CP(label, int, 3) CP(frame, u32, 1) CP(node_id, int, 3)
#undef CP
  // (wait = false: every field went through the page-locked staging — at most 256 KB each — or is read from the caller's
  // memory until the caller's next wait on the stream: sgtd_search_frame, whose inputs stay valid for the call)
  if (wait) CHK(xfer_sync(e));
  return SGTD_OK;
}

void note_frames(sgtd_engine *e, u32 lo, u32 hi) {
  if (!e->have_frames) {
    e->frame_lo = lo; e->frame_hi = hi; e->have_frames = true;
  } else {
    e->frame_lo = std::min(e->frame_lo, lo);
    e->frame_hi = std::max(e->frame_hi, hi);
  }
}

// ---------------------------------------------------------------------------
// finalize: sort + CSR + hash
// ---------------------------------------------------------------------------
// stable LSD radix sort of (key, val) pairs by the low `bits` of the key, 8-bit digits; passes
// whose digit is the same for every key are skipped (skip_const: costs a host round trip per
// pass, only used by the once-per-map table build).  The result is in kin / vin.
template <class KeyT>
int radix_sort_pairs(sgtd_engine *e, KeyT *&kin, KeyT *&kout, u32 *&vin, u32 *&vout, long long n, int bits,
                     bool skip_const, const u32 *n_dev = nullptr) {
  const int nblocks = (int)((n + SGTD_RS_TILE - 1) / SGTD_RS_TILE);
  CHK(ensure(e, e->hist, (size_t)256 * nblocks * sizeof(u32)));
  CHK(ensure(e, e->digit_tot, 256 * sizeof(u32)));
  std::vector<u32> tot(256);
  for (int shift = 0; shift < bits; shift += 8) {
    u32 *hist = e->hist.as<u32>();
    radix_hist_kernel<KeyT><<<nblocks, SGTD_RS_THREADS, 0, e->stream>>>(kin, n, shift, hist, nblocks, n_dev);
    HIPCHK(hipGetLastError());
    if (skip_const) {
      radix_digit_totals_kernel<<<256, SGTD_SCAN_THREADS, 0, e->stream>>>(hist, nblocks, e->digit_tot.as<u32>());
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(tot.data(), e->digit_tot.p, 256 * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      bool constant = false;
      for (int d = 0; d < 256; d++)
        if ((long long)tot[d] == n) constant = true;
      if (constant) continue;  // every key has the same digit here: the pass is the identity
    }
    CHK(device_scan(e, hist, hist, (long long)256 * nblocks));
    radix_scatter_kernel<KeyT><<<nblocks, SGTD_RS_THREADS, 0, e->stream>>>(kin, vin, kout, vout, n, shift,
                                                                       e->hist.as<u32>(), nblocks, n_dev);
    HIPCHK(hipGetLastError());
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  return SGTD_OK;
}

IdMap id_map(const sgtd_engine *e, u32 bits) {
  IdMap m;
  m.frame_first = e->frame_first.as<u32>();
  m.by_frame = e->id_by_frame ? e->by_frame.as<u32>() : nullptr;
  m.bits = bits;
  m.frame_lo = e->have_frames ? e->frame_lo : 0;
  return m;
}

// probe layout of the entries [g0, g1) into segment S: sort by key, slices, directory, hash
int build_segment(sgtd_engine *e, sgtd_engine::Segment &S, long long g0, long long g1) {
  const long long E = g1 - g0;
  S.g0 = g0; S.g1 = g1;
  S.n_buckets = 0;
  S.hash_mask = 1023;
  S.sum_len_sq = 0.0;
  S.frame_hi = e->have_frames ? e->frame_hi : 0;
  S.built = true;
  if (E == 0) {
    CHK(ensure(e, S.hash, (size_t)1024 * sizeof(HashSlot)));
    HIPCHK(hipMemsetAsync(S.hash.p, 0xFF, (size_t)1024 * sizeof(HashSlot), e->stream));
    return SGTD_OK;
  }
  // the arrays of this range, indexed by the entry's position inside it
  const double *side = e->tab.side.as<double>() + (size_t)g0 * 3;
  const int *label = e->tab.label.as<int>() + (size_t)g0 * 3;
  const u32 *frame = e->tab.frame.as<u32>() + (size_t)g0;
  CHK(ensure(e, e->keyA, (size_t)E * sizeof(u64)));
  CHK(ensure(e, e->keyB, (size_t)E * sizeof(u64)));
  CHK(ensure(e, e->valA, (size_t)E * sizeof(u32)));
  CHK(ensure(e, e->valB, (size_t)E * sizeof(u32)));
  CHK(ensure(e, e->bad_flag, 2 * sizeof(int)));
  HIPCHK(hipMemsetAsync(e->bad_flag.p, 0, 2 * sizeof(int), e->stream));
  make_keys_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(side, label, e->keyA.as<u64>(), e->valA.as<u32>(), E,
                                                             e->bad_flag.as<int>());
  HIPCHK(hipGetLastError());
  frame_monotone_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(frame, E, e->bad_flag.as<int>() + 1);
  HIPCHK(hipGetLastError());
  int bad[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(bad, e->bad_flag.p, sizeof(bad), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (bad[0]) { S.built = false; return SGTD_ERR_UNSUPPORTED; }  // a cell coordinate beyond 16 bits
  const bool monotone = bad[1] == 0;

  u64 *kin = e->keyA.as<u64>(), *kout = e->keyB.as<u64>();
  u32 *vin = e->valA.as<u32>(), *vout = e->valB.as<u32>();
  CHK(ensure(e, e->slice_of, (size_t)E));
  if (!monotone) {
    // frame ids out of insertion order (caller-stamped descriptors): the slice assignment needs
    // every bucket grouped by frame, so sort by (key, frame, g) for it, then start over
    frame_keys_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(frame, kin, vin, E);
    HIPCHK(hipGetLastError());
    CHK(radix_sort_pairs(e, kin, kout, vin, vout, E, 32, true));
    keys_of_order_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(side, label, vin, kin, E);
    HIPCHK(hipGetLastError());
    CHK(radix_sort_pairs(e, kin, kout, vin, vout, E, 60, true));
    slice_assign_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(kin, vin, side, frame, E, e->dc.rough, e->slice_of.as<unsigned char>());
    HIPCHK(hipGetLastError());
    kin = e->keyA.as<u64>(); kout = e->keyB.as<u64>(); vin = e->valA.as<u32>(); vout = e->valB.as<u32>();
    make_keys_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(side, label, kin, vin, E, e->bad_flag.as<int>());
    HIPCHK(hipGetLastError());
  }
  CHK(radix_sort_pairs(e, kin, kout, vin, vout, E, 60, true));   // (key, g): buckets in insertion order
  if (monotone) {
    slice_assign_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(kin, vin, side, frame, E, e->dc.rough, e->slice_of.as<unsigned char>());
    HIPCHK(hipGetLastError());
  }
  // buckets
  CHK(ensure(e, e->flags, (size_t)E * sizeof(u32)));
  head_flags_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(kin, e->flags.as<u32>(), E);
  HIPCHK(hipGetLastError());
  u32 last_flag = 0, last_excl = 0;
  HIPCHK(hipMemcpyAsync(&last_flag, e->flags.as<u32>() + (E - 1), sizeof(u32), hipMemcpyDeviceToHost, e->stream));
  CHK(device_scan(e, e->flags.as<u32>(), e->flags.as<u32>(), E));
  HIPCHK(hipMemcpyAsync(&last_excl, e->flags.as<u32>() + (E - 1), sizeof(u32), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  const u32 U = last_excl + last_flag;
  S.n_buckets = U;
  CHK(ensure(e, S.bucket_start, (size_t)(U + 1) * sizeof(u32)));
  CHK(ensure(e, S.bucket_key, (size_t)U * sizeof(u64)));
  bucket_starts_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(kin, e->flags.as<u32>(), S.bucket_start.as<u32>(),
                                                                 S.bucket_key.as<u64>(), E);
  HIPCHK(hipGetLastError());
  // probe order: every bucket partitioned by slice, insertion order inside a slice
  CHK(ensure(e, S.perm, (size_t)(E + SGTD_SENTINELS) * sizeof(u32)));
  CHK(ensure(e, S.dir, (size_t)U * sizeof(BucketDir)));
  slice_partition_kernel<<<e->n_cus * 8, 256, 0, e->stream>>>(S.bucket_start.as<u32>(), U, (u32)E, vin,
                                                               e->slice_of.as<unsigned char>(), S.perm.as<u32>(),
                                                               S.dir.as<BucketDir>());
  HIPCHK(hipGetLastError());
  // the probe layout of the tail segment lies BEHIND the main segment's (and its sentinels) in the main
  // segment's buffer: one base address for both, so that a pass's visit list can run through both segments
  HotEntry *hot = nullptr;
  if (&S == &e->seg[1]) {
    sgtd_engine::Segment &M = e->seg[0];
    const size_t m_ent = (size_t)(M.g1 - M.g0) + SGTD_SENTINELS;
    CHK(ensure(e, M.hot, (m_ent + (size_t)E + SGTD_SENTINELS) * sizeof(HotEntry), /*keep=*/true));
    hot = M.hot.as<HotEntry>() + m_ent;
  } else {
    // (with room for the largest tail do_finalize accepts, so that an append does not move the layout;
    // none under the SGTD_TAIL_MAX test hook: the tail's build then grows the buffer and moves it)
    const size_t tail_room = e->tail_max > 0 ? 0 : (size_t)std::max<long long>(262144, E / 8) + SGTD_SENTINELS;
    CHK(ensure(e, S.hot, ((size_t)(E + SGTD_SENTINELS) + tail_room) * sizeof(HotEntry)));
    hot = S.hot.as<HotEntry>();
  }
  gather_hot_kernel<<<grid_for(E + SGTD_SENTINELS, 256), 256, 0, e->stream>>>(
      S.perm.as<u32>(), side, frame, e->id_by_frame ? e->id_of_g.as<u32>() : nullptr, id_map(e, e->id_bits), hot, E, (u32)g0);
  HIPCHK(hipGetLastError());
  u32 cap = 1024;
  while (cap < 2ull * U) cap <<= 1;
  S.hash_mask = cap - 1;
  CHK(ensure(e, S.hash, (size_t)cap * sizeof(HashSlot)));
  HIPCHK(hipMemsetAsync(S.hash.p, 0xFF, (size_t)cap * sizeof(HashSlot), e->stream));
  hash_insert_kernel<<<grid_for(U, 256), 256, 0, e->stream>>>(S.bucket_key.as<u64>(), S.bucket_start.as<u32>(), U,
                                                               (u32)E, S.hash.as<HashSlot>(), S.hash_mask);
  HIPCHK(hipGetLastError());
  CHK(ensure(e, e->sq_sum, sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(e->sq_sum.p, 0, sizeof(unsigned long long), e->stream));
  bucket_sq_kernel<<<grid_for(U, 256), 256, 0, e->stream>>>(S.bucket_start.as<u32>(), U, (u32)E, e->sq_sum.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  unsigned long long sq = 0;
  HIPCHK(hipMemcpyAsync(&sq, e->sq_sum.p, sizeof(sq), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  S.sum_len_sq = (double)sq;
  return SGTD_OK;
}

// The entry ids (IdMap) of the whole table: frame_first (+ by_frame, id_of_g when frame ids are out
// of insertion order) and the number of bits the in-frame rank needs.  `bits` is what the ids of
// the segments built from now on use; a change invalidates segments built with another value.
int build_idmap(sgtd_engine *e, u32 &bits) {
  const long long E = e->n_entries;
  bits = e->id_bits ? e->id_bits : 13;
  if (E == 0 || !e->have_frames) return SGTD_OK;
  const u32 lo = e->frame_lo, span = e->frame_hi - e->frame_lo + 1;
  const u32 *frame = e->tab.frame.as<u32>();
  CHK(ensure(e, e->frame_first, (size_t)span * sizeof(u32)));
  CHK(ensure(e, e->longest, 2 * sizeof(u32)));
  HIPCHK(hipMemsetAsync(e->longest.p, 0, 2 * sizeof(u32), e->stream));
  frame_monotone_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(frame, E, reinterpret_cast<int *>(e->longest.as<u32>() + 1));
  HIPCHK(hipGetLastError());
  u32 h[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h, e->longest.p, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->id_by_frame = h[1] != 0;
  const u32 *key = frame;
  if (e->id_by_frame) {
    CHK(ensure(e, e->keyA, (size_t)E * sizeof(u64)));
    CHK(ensure(e, e->keyB, (size_t)E * sizeof(u64)));
    CHK(ensure(e, e->valA, (size_t)E * sizeof(u32)));
    CHK(ensure(e, e->valB, (size_t)E * sizeof(u32)));
    u64 *kin = e->keyA.as<u64>(), *kout = e->keyB.as<u64>();
    u32 *vin = e->valA.as<u32>(), *vout = e->valB.as<u32>();
    frame_keys_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(frame, kin, vin, E);
    HIPCHK(hipGetLastError());
    CHK(radix_sort_pairs(e, kin, kout, vin, vout, E, 32, true));     // stable: insertion order inside a frame
    CHK(ensure(e, e->by_frame, (size_t)E * sizeof(u32)));
    CHK(ensure(e, e->id_of_g, (size_t)E * sizeof(u32)));
    HIPCHK(hipMemcpyAsync(e->by_frame.p, vin, (size_t)E * sizeof(u32), hipMemcpyDeviceToDevice, e->stream));
    // the sorted frame ids as 32-bit words in the sort's spare value buffer (the sort is over; the
    // kernels below run before build_segment reuses the scratch)
    low_words_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(kin, E, vout);
    HIPCHK(hipGetLastError());
    key = vout;
  }
  frame_first_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(key, E, lo, e->frame_first.as<u32>());
  HIPCHK(hipGetLastError());
  frame_longest_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(key, E, lo, e->frame_first.as<u32>(), e->longest.as<u32>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h, e->longest.p, sizeof(u32), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  u32 need = 1;
  while (need < 32 && (1ull << need) < (unsigned long long)h[0]) need++;
  // the local frame 2^(32 - bits) - 1 is the dead record's: span must stay below it
  auto fits = [&](u32 b) { return b < 32 && (unsigned long long)span < (1ull << (32 - b)) - 1ull; };
  if (bits < need || !fits(bits)) {
    bits = std::max<u32>(need, 13);
    if (!fits(bits)) bits = need;
    if (!fits(bits)) {
      e->err = "a 32-bit entry id cannot hold this table's frame span and its largest frame";
      return SGTD_ERR_UNSUPPORTED;
    }
  }
  if (e->id_by_frame) {
    id_of_sorted_kernel<<<grid_for(E, 256), 256, 0, e->stream>>>(key, e->by_frame.as<u32>(), E, lo, e->frame_first.as<u32>(), bits,
                                                                 e->id_of_g.as<u32>());
    HIPCHK(hipGetLastError());
  }
  return SGTD_OK;
}

// AddSTDescs only ever appends (STDesc.cpp:149-172).  After an append to a finalized table only
// the appended entries are sorted, into the tail segment (cost proportional to the tail); the
// query sweeps both segments — inside a bucket the reference's order is insertion order, i.e.
// main entries before tail entries, and a frame's entries all live in ONE segment (the tail only
// takes frames newer than every frame of the main segment), so the per-frame match order is
// unchanged.  The tail is merged (one full build) when it outgrows an eighth of the main
// segment; force_merge asks for the single-segment form (table dump).
int do_finalize(sgtd_engine *e, bool force_merge = false) {
  if (e->finalized && !(force_merge && e->n_seg > 1)) return SGTD_OK;
  if (e->attached_to) { e->err = "the table belongs to another handle (sgtd_attach_table): only its owner rebuilds it"; return SGTD_ERR_STATE; }
  e->table_version++;
  const long long E = e->n_entries;
  if (E >= (1ll << 32) - 2 - SGTD_SENTINELS) return SGTD_ERR_UNSUPPORTED;
  const auto t0 = std::chrono::steady_clock::now();
  sgtd_engine::Segment &M = e->seg[0], &T = e->seg[1];
  const long long tail_max = e->tail_max > 0 ? (long long)e->tail_max : std::max<long long>(262144, (M.g1 - M.g0) / 8);
  u32 bits = 0;
  CHK(build_idmap(e, bits));
  // (an appended frame older than the main segment's newest, a frame that outgrew the ids' rank
  // bits, or frame ids that just went out of insertion order: everything is rebuilt)
  const bool can_tail = !force_merge && M.built && M.g0 == 0 && M.g1 > 0 && M.g1 <= E && E - M.g1 <= tail_max &&
                        bits == e->id_bits && !e->id_by_frame && (E == M.g1 || e->append_min_frame > M.frame_hi);
  e->id_bits = bits;
  if (can_tail) {
    if (E > M.g1) { CHK(build_segment(e, T, M.g1, E)); e->n_seg = 2; }
    else { T.built = false; e->n_seg = 1; }
  } else {
    CHK(build_segment(e, M, 0, E));
    T.built = false; T.g0 = T.g1 = E;
    e->n_seg = 1;
    e->append_min_frame = 0xFFFFFFFFu;
  }
  e->ms_finalize = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  e->finalized = true;
  return SGTD_OK;
}

// ---------------------------------------------------------------------------
// the query pipeline on descriptors already in e->qd (strided)
// ---------------------------------------------------------------------------
int rec_alloc(sgtd_engine *e, bool compact_lists) {
#ifdef SGTD_EXP_SHADOW16
  CHK(ensure(e, e->rec, (e->rec_cap + 16) * sizeof(u32) * 3 / 2 + 256));     // (experiment build: the 2-byte shadow stream behind the records)
#else
  CHK(ensure(e, e->rec, (e->rec_cap + 16) * sizeof(u32)));     // + a few quads: the list passes read four records at the last list's tail
#endif
  // (the compact candidate lists between block_count and block_write: every block reserves room for all of its records —
  // twice the record buffer; the per-query list pass needs none)
  if (compact_lists) CHK(ensure(e, e->c_pair, std::min(e->rec_cap, kIndexLimit) * sizeof(u64)));
  if (e->diag) {
    CHK(ensure(e, e->rec_cell, e->rec_cap));
    CHK(ensure(e, e->rec_dis, e->rec_cap * sizeof(double)));
  }
  CHK(ensure(e, e->pairs, e->pair_cap * sizeof(u64)));
  CHK(ensure(e, e->amb_queue, std::max<size_t>(65536, e->rec_cap / 64) * sizeof(uint2)));
  return SGTD_OK;
}

// descriptors per block of the block passes over the match records (one wave per block): 128 — or 64 / 32 where the batch would
// otherwise have fewer than a thousand blocks (a one-frame batch: 57 blocks = 57 waves on 1 024 SIMDs; SGTD_BLOCK_CHUNK overrides)
u32 block_chunk(const sgtd_engine *e) {
  if (const char *o = getenv("SGTD_BLOCK_CHUNK")) { const int c = atoi(o); if (c == 32 || c == 64 || c == 128) return (u32)c; }
  u32 c = SGTD_PROBE_CHUNK;
  while (c > SGTD_SUB_DESCS && (long long)e->nq * ((e->q_stride + c - 1) / c) < 1024) c >>= 1;
  return c;
}

struct Views {
  TableView T;
  QueryView Q;
  ProbeBuffers B;
  u32 span;
  int blocks_per_query;
};

Views make_views(sgtd_engine *e) {
  Views v;
  v.span = e->have_frames ? (e->frame_hi - e->frame_lo + 1) : 1;
  TableView &T = v.T;
  const sgtd_engine::Segment &S = e->seg[0], &S1 = e->seg[1];
  T.ent = S.hot.as<HotEntry>(); T.map = id_map(e, e->id_bits ? e->id_bits : 13); T.cold_side = e->tab.side.as<double>();
  T.dir = S.dir.as<BucketDir>();
  T.hash = S.hash.as<HashSlot>(); T.hash_mask = S.hash_mask;
  T.coarse_at = e->coarse_at; T.whole_at = e->whole_at;
  T.n_entries = (u32)(S.g1 - S.g0); T.frame_lo = e->have_frames ? e->frame_lo : 0; T.frame_span = v.span;
  // the tail segment (appends after the table was finalized): its own directory and hash, its probe
  // layout behind the main segment's in the same buffer
  T.tail_off = e->n_seg > 1 ? T.n_entries + SGTD_SENTINELS : 0u;
  T.dir1 = S1.dir.as<BucketDir>(); T.hash1 = S1.hash.as<HashSlot>(); T.hash_mask1 = S1.hash_mask;
  T.n_entries1 = e->n_seg > 1 ? (u32)(S1.g1 - S1.g0) : 0u;
  QueryView &Q = v.Q;
  Q.side = e->qd.side.as<double>(); Q.qrec = e->qd.qrec.as<QueryRec>();
  Q.label = e->qd.label.as<int>(); Q.frame = e->qd.frame.as<u32>();
  Q.count = e->q_count.as<u32>(); Q.stride = e->q_stride; Q.n_queries = e->nq; Q.chunk = block_chunk(e);
  ProbeBuffers &B = v.B;
  B.rec_cell = e->rec_cell.as<unsigned char>(); B.rec_dis = e->rec_dis.as<double>();
  B.rec_cap = (u32)std::min<size_t>(e->rec_cap >> SGTD_REC_SHIFT, kIndexLimit);      // granules
  B.rec = e->rec.as<u32>(); B.id_bits = e->id_bits ? e->id_bits : 13;
  B.ctr = e->cursors.as<u32>();
  // the smallest slab: the streams of all resident waves hold one each (8192 waves x SGTD_PAIR streams) — together
  // at most a quarter of the record buffer (a 33 M-record buffer of a small batch: 512-record slabs).  (A slab is one
  // atomic add to the cursor, and ONE address takes 88 M of them per second, tools/atomic_rate.hip: the 1.8 G records
  // of the default batch in slabs of 8 K are 2.4 ms of the cursor's time inside a 5 ms sweep — but slabs of 32 K or
  // 128 K run no faster, SGTD_REC_SLAB: the waves do not wait for it.)
  {
    const size_t streams = (size_t)e->n_cus * 32 * SGTD_PAIR;
    u32 slab = SGTD_REC_SLAB;
    while (slab > 512u && (size_t)slab * streams * 4 > e->rec_cap) slab >>= 1;
    if (const char *o = getenv("SGTD_REC_SLAB")) slab = (u32)std::max(512, atoi(o));   // experiment knob
    B.rec_slab = slab >> SGTD_REC_SHIFT;      // granules
  }
  // room a list gets when its pass starts (ProbeBuffers::rec_rate): three times the matches per visited entry
  // and descriptor of the batch before; a quarter of the visit list for the first batch
  {
    const double per_pass = SGTD_PAIR >= 4 ? 3.2 : (SGTD_PAIR >= 2 ? 1.9 : 1.0);
    B.rec_rate = 64;
    if (e->stats.last_P_swept > 0 && e->stats.last_M > 0) {
      // three times the batch before's matches per visited entry and descriptor — but not more than lets the
      // reservations of a batch like the one before (rate / 256 of every pass's visit list per descriptor, + 256
      // records each) fit three fifths of the record buffer (slabs strand an eighth, batches differ): on maps whose visit lists match little (skewed label
      // frequencies: a twelfth of the visits) three times the rate over-reserves the 32-bit record index by itself,
      // and every batch would pay a re-run.  Never below 1.2 times the measured rate (lists that outgrow their room move).
      const double meas = 256.0 * (double)e->stats.last_M / ((double)e->stats.last_P_swept * per_pass);
      const double scale = e->stats.last_queries > 0 ? (double)e->nq / (double)e->stats.last_queries : 1.0;
      const double fit = 256.0 * (0.6 * (double)e->rec_cap - 256.0 * (double)e->stats.last_D * scale) /
                         std::max(1.0, (double)e->stats.last_P_swept * scale * per_pass);
      const double want = std::max(std::min(3.0 * meas, fit), 1.2 * meas);
      B.rec_rate = (u32)std::min(256.0, std::max(want < 16.0 && fit < 16.0 ? 2.0 : 16.0, std::ceil(want)));
    }
    B.rec_rate = std::min(B.rec_rate, e->rec_rate_cap);
    if (e->diag) B.rec_rate = 256;
    if (e->rec_rate_hook) B.rec_rate = e->rec_rate_hook;
  }
  B.amb_queue = e->amb_queue.as<uint2>();
  B.amb_cap = (u32)std::min<size_t>(e->amb_queue.bytes / sizeof(uint2), 0xFFFFFFF0u);
  B.list = e->list.as<uint2>(); B.n_visit = e->n_visit.as<u32>();
  B.votes = e->votes.as<u32>();
  v.blocks_per_query = (int)((e->q_stride + block_chunk(e) - 1) / block_chunk(e));
  return v;
}

// pass 2 of the match-list assembly (both word widths of the compact lists)
int launch_block_write(sgtd_engine *e, const Views &v, const CompactLists &CL, int agrid, int blocks) {
  const int cn = e->dc.cand_num;
  if (v.T.map.bits <= SGTD_NARROW_RANK_BITS && !e->wide_pairs)
    block_write_kernel<true><<<agrid, 256, 0, e->stream>>>(v.Q, v.B, CL, blocks, e->blk_count.as<u32>(), cn, e->pair_off.as<long long>(),
                                                            e->q_pair_base.as<u32>(), e->pairs.as<u64>(), v.T.map, e->n_cand.as<int>(),
                                                            e->cand_frame.as<int>());
  else
    block_write_kernel<false><<<agrid, 256, 0, e->stream>>>(v.Q, v.B, CL, blocks, e->blk_count.as<u32>(), cn, e->pair_off.as<long long>(),
                                                             e->q_pair_base.as<u32>(), e->pairs.as<u64>(), v.T.map, e->n_cand.as<int>(),
                                                             e->cand_frame.as<int>());
  HIPCHK(hipGetLastError());
  return SGTD_OK;
}

// the match lists of every query by a workgroup of its own (select_kernels.hip.h); the slot of a record's frame
// from a byte table in LDS while the frame span fits, else from the hash of the candidates
int launch_pairs_query(sgtd_engine *e, const Views &v, const u64 *keep = nullptr) {
  const int cn = e->dc.cand_num;
  const size_t img = (size_t)SGTD_PQ_TILE_RECS * sizeof(u32);
  const size_t tab = (((size_t)v.span + 15) & ~(size_t)15) + 16;     // (+ the bytes that answer for dead records)
  // (up to 100 KB two workgroups share a CU; a span whose byte table only fits alone — 100 000 frames: 132 KB with the
  // image — still beats the candidates' hash: one workgroup per CU)
  static const size_t room = lds_room(&pairs_query_kernel<true>);
  if (img + tab <= room) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pairs_query_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(img + tab)));
    pairs_query_kernel<true><<<e->nq, SGTD_PQ_THREADS, img + tab, e->stream>>>(v.Q, v.B, e->n_cand.as<int>(), e->cand_frame.as<int>(), cn,
                                                                                e->pair_off.as<long long>(), e->q_pair_base.as<u32>(),
                                                                                e->pairs.as<u64>(), v.T.map, v.span, v.T.frame_lo, keep);
  } else {
    pairs_query_kernel<false><<<e->nq, SGTD_PQ_THREADS, img, e->stream>>>(v.Q, v.B, e->n_cand.as<int>(), e->cand_frame.as<int>(), cn,
                                                                           e->pair_off.as<long long>(), e->q_pair_base.as<u32>(),
                                                                           e->pairs.as<u64>(), v.T.map, v.span, v.T.frame_lo, keep);
  }
  HIPCHK(hipGetLastError());
  return SGTD_OK;
}

// the batch's local candidate tables, packed, into the buffer a multi-GPU caller registered (exchange_kernels.hip.h);
// enqueued as soon as they are final, before the match lists are written
int export_candidates(sgtd_engine *e, const Views &v) {
  if (!e->export_packed) return SGTD_OK;
  const int nq = e->nq, cn = e->dc.cand_num;
  if (xchg_packed_ints(nq, cn) > e->export_cap) {
    e->err = "the registered candidate-export buffer is too small for this batch";
    return SGTD_ERR_CAPACITY;
  }
  if (e->export_busy) {     // a side stream may still be reading the table of the batch before
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_export_free, 0));
    e->export_busy = false;
  }
  export_candidates_kernel<<<std::min(grid_for((long long)nq * cn, 256), 512), 256, 0, e->stream>>>(
      e->cand_frame.as<int>(), e->cand_votes.as<int>(), v.B.overflow(), nq, cn, ++e->batch_serial, e->export_packed);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev_cand, e->stream));
  return SGTD_OK;
}

// The match lists of a batch whose candidate tables are final, by one workgroup per query (select_kernels.hip.h):
// offsets (the prefix sums of the candidates' votes — of the candidates in `keep` only, when a mask is given), the
// queries' bases, the lists.  Runs once per batch inside launch_select, or later and repeatedly through
// sgtd_finish_lists (the records stay intact: a list pass can be repeated with another mask).
int launch_lists(sgtd_engine *e, const Views &v, const u64 *keep, bool first) {
  const int nq = e->nq, cn = e->dc.cand_num;
  if (keep) {
    cand_prefix_masked_kernel<<<grid_for(nq, 256), 256, 0, e->stream>>>(e->n_cand.as<int>(), e->cand_votes.as<int>(), keep, cn, nq,
                                                                          e->pair_off.as<long long>(), e->q_pairs.as<u32>());
    HIPCHK(hipGetLastError());
  } else if (!e->fused_votes_last || e->lists_masked) {
    // (votes_topk_kernel leaves the unmasked offsets itself)
    cand_prefix_kernel<<<grid_for(nq, 256), 256, 0, e->stream>>>(e->n_cand.as<int>(), e->cand_votes.as<int>(), cn, nq,
                                                                   e->pair_off.as<long long>(), e->q_pairs.as<u32>());
    HIPCHK(hipGetLastError());
  }
  e->lists_masked = keep != nullptr;
  e->list_keep = keep;
  if (e->timing && first) HIPCHK(hipEventRecord(e->ev[EV_COUNT_T], e->stream));
  if (!first) HIPCHK(hipMemsetAsync(v.B.overflow() + 1, 0, sizeof(int), e->stream));
  query_base_kernel<<<1, 256, 0, e->stream>>>(e->q_pairs.as<u32>(), e->q_pair_base.as<u32>(), nq,
                                               (u32)std::min<size_t>(e->pair_cap, 0xFFFFFFF0u), v.B.overflow());
  HIPCHK(hipGetLastError());
  if (e->timing && first) HIPCHK(hipEventRecord(e->ev[EV_SCAN], e->stream));
  CHK(launch_pairs_query(e, v, keep));
  if (e->timing && first) HIPCHK(hipEventRecord(e->ev[EV_WRITE], e->stream));
  if (first) {
    batch_totals_kernel<<<1, 1, 0, e->stream>>>(v.B.ctr, e->totals.as<unsigned long long>());
    HIPCHK(hipGetLastError());
  }
  e->lists_pending = false;
  e->verified = false;
  e->batch_synced = false;
  return SGTD_OK;
}

int launch_select(sgtd_engine *e) {
  const int nq = e->nq;
  const long long n_slots = (long long)nq * e->q_stride;
  const int cn = e->dc.cand_num;
  const u32 span = e->have_frames ? (e->frame_hi - e->frame_lo + 1) : 1;
  const int blocks = (int)((e->q_stride + block_chunk(e) - 1) / block_chunk(e));
  CHK(ensure(e, e->cursors, kCtrWords * sizeof(u32)));
  if (!e->totals.p) {
    CHK(ensure(e, e->totals, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(e->totals.p, 0, 4 * sizeof(unsigned long long), e->stream));
  }
  CHK(ensure(e, e->list, (size_t)std::max<long long>(n_slots, 1) * sizeof(uint2)));
  CHK(ensure(e, e->n_visit, (size_t)std::max<long long>(n_slots, 1) * sizeof(u32)));
  CHK(ensure(e, e->votes, (size_t)nq * span * sizeof(u32)));
  CHK(ensure(e, e->slot_of, (size_t)nq * span + 4));      // (+ a word: small_order_kernel clears it by words)
  CHK(ensure(e, e->q_M, (size_t)nq * sizeof(u32)));
  CHK(ensure(e, e->q_P, (size_t)nq * sizeof(unsigned long long)));
  CHK(ensure(e, e->q_pairs, (size_t)nq * sizeof(u32)));
  CHK(ensure(e, e->q_pair_base, (size_t)(nq + 1) * sizeof(u32)));
  CHK(ensure(e, e->blk_count, (size_t)nq * blocks * 64 * sizeof(u32)));
  CHK(ensure(e, e->c_blk, (size_t)nq * blocks * 2 * sizeof(u32)));
  CHK(ensure(e, e->n_cand, (size_t)nq * sizeof(int)));
  CHK(ensure(e, e->cand_frame, (size_t)nq * cn * sizeof(int)));
  CHK(ensure(e, e->cand_votes, (size_t)nq * cn * sizeof(int)));
  CHK(ensure(e, e->pair_off, (size_t)nq * (cn + 1) * sizeof(long long)));

  // Which passes over the match records (STDesc.cpp:404-453): one workgroup per query (select_kernels.hip.h) when
  // the batch has a query for every CU — votes + top-k in one launch while the query's vote histogram fits LDS,
  // the match lists in one launch while an entry's rank among its frame's fits the image word — else the
  // five-kernel form with one wave per 128-descriptor block.
  const u32 tile_span = span <= 36 * 1024 ? span : 36 * 1024;
  const u32 n_tiles = (span + tile_span - 1) / tile_span;
  // (automatic choice where a query's vote histogram fits LDS: with the candidates' hash instead of the frame -> slot
  // byte table the list pass is slower than the block passes — 100 000-frame map, 256 queries: 6.1 against 2.9 ms)
  static const size_t votes_room = lds_room(&votes_topk_kernel), list_room = lds_room(&pairs_query_kernel<true>);
  const bool votes_fit = votes_topk_lds_bytes(span) <= votes_room;
  // ... or at least its frame -> slot byte table beside a tile's image (one workgroup per CU then; 100 000 frames, 256 queries:
  // votes by tiles + top-k + this list pass 15.1 ms per step against 16.1 with the block passes, gpurun_out/r05o_*)
  const bool table_fits = (size_t)SGTD_PQ_TILE_RECS * sizeof(u32) + (((size_t)span + 15) & ~(size_t)15) + 16 <= list_room;
  const bool per_query = e->select_mode == 2 || (e->select_mode == 0 && nq >= e->n_cus && (votes_fit || table_fits));
  // (an image word of the list pass is slot(6) | descriptor(9) | rank: with 64 candidates AND ranks of the full width the
  // word of slot 63, descriptor 511, rank 2^17 - 1 would be the pass's "no record" marker — that corner takes the block form)
  const bool fused_pairs = per_query && !e->wide_pairs && (e->id_bits ? e->id_bits : 13) <= SGTD_PQ_RANK_BITS &&
                           !(cn == SGTD_MAX_CAND && (e->id_bits ? e->id_bits : 13) == SGTD_PQ_RANK_BITS);
  const bool fused_votes = fused_pairs && votes_fit;     // (block_count_kernel wants topk_kernel's slot table)
  CHK(rec_alloc(e, !fused_pairs));
  const bool votes_per_query = n_tiles == 1 ? nq >= e->n_cus : ((long long)nq * n_tiles >= e->n_cus / 4 && n_tiles <= 8);
  // bits per cell coordinate of the sort key: the largest cell a built descriptor can have
  int cbits = 1;
  while ((1ll << cbits) < (long long)(e->dc.max_len * e->dc.scale) + 3 && cbits < 16) cbits++;
  // (+ the position inside the cell, home_keys_kernel: the bits that are free below the next multiple of a sort
  // digit, or four bits and one more pass)
  const int spare = (8 - (12 + 3 * cbits) % 8) % 8;
  int sub_bits = spare >= 2 ? std::min(spare, 6) : 4;
  if (const char *o = getenv("SGTD_HOME_SUB_BITS")) sub_bits = std::min(6, std::max(0, atoi(o)));   // experiment knob
  const int key_bits = 12 + 3 * cbits + sub_bits;
  // one frame per call: the clearing of the batch's counters and tables and the whole ordering of its descriptors in ONE
  // launch (small_order_kernel; SGTD_SMALL_ORDER=0: the general form)
  const bool small_on = [] { const char *o = getenv("SGTD_SMALL_ORDER"); return !(o && !atoi(o)); }();
  const bool small = small_on && nq == 1 && n_slots <= SGTD_SMALL_SLOTS && key_bits <= 32 && !fused_votes && span <= (1u << 20);
  if (e->thr2_pending > 0) {      // (not inside small_order_kernel: the thresholds and gate masks of 7 000 descriptors are 50 us of ONE workgroup's time, 5 us of eighteen's)
    thr2_kernel<<<grid_for(e->thr2_pending, 256), 256, 0, e->stream>>>(e->qd.side.as<double>(), e->qd.frame.as<u32>(), e->qd.qrec.as<QueryRec>(), e->thr2_pending,
                                                                        e->dc.rough);
    HIPCHK(hipGetLastError());
  }
  if (!small) {
    HIPCHK(hipMemsetAsync(e->cursors.p, 0, kCtrWords * sizeof(u32), e->stream));
    if (!fused_votes) {
      if (!votes_per_query) HIPCHK(hipMemsetAsync(e->votes.p, 0, (size_t)nq * span * sizeof(u32), e->stream));   // (global vote atomics)
      HIPCHK(hipMemsetAsync(e->slot_of.p, 0xFF, (size_t)nq * span, e->stream));
      HIPCHK(hipMemsetAsync(e->cand_frame.p, 0xFF, (size_t)nq * cn * sizeof(int), e->stream));
      HIPCHK(hipMemsetAsync(e->cand_votes.p, 0, (size_t)nq * cn * sizeof(int), e->stream));
    }
    HIPCHK(hipMemsetAsync(e->q_M.p, 0, (size_t)nq * sizeof(u32), e->stream));
    HIPCHK(hipMemsetAsync(e->q_P.p, 0, (size_t)nq * sizeof(unsigned long long), e->stream));
  }

  Views v = make_views(e);
  const size_t hist_bytes = (size_t)span * sizeof(u32);
  const bool lds_votes = hist_bytes <= 150 * 1024;
  const int groups = (blocks + 3) / 4;
  const int agrid = ((nq + 7) / 8) * groups * 8;   // workgroup b serves query (b/8/groups)*8 + b%8
  // one GroupRow (1 KB) per distinct home cell of the batch: at most one per descriptor slot, in practice a
  // twelfth of that (0.7 M cells for 9.1 M descriptors at the default batch).  Reserved: every slot for
  // small batches, an eighth of the slots for large ones — a 15-GB reservation cost the first batch
  // 0.4 s of hipMalloc — and what a batch really needed, plus a quarter, once one has overflowed it
  // (group_resolve_kernel raises the overflow flag, sync_batch re-runs).
  if (e->group_cap_hook) { if (e->group_cap == 0) e->group_cap = e->group_cap_hook; }
  else {
    // distinct home cells of a batch: every slot's for small batches (a single frame: ~40 % of its slots), about
    // 1.5 M at most on the shipped resolution (0.39 M for 256 frames, 0.7 M for 2048), never more than half the slots
    const long long want = n_slots <= 65536 ? n_slots : std::min<long long>(n_slots / 2, std::max<long long>(n_slots / 8, 1500000));
    e->group_cap = std::max<size_t>(e->group_cap, (size_t)want);
  }
  const int row_slots = e->n_seg > 1 ? 2 : 1;       // with a tail segment: its 27 rows in the second KB of the slot
  CHK(ensure(e, e->cell_rows, std::max<size_t>(e->group_cap, 1) * SGTD_GROUP_ROW_BYTES * row_slots));
  const unsigned char *rows = e->cell_rows.as<unsigned char>();
  const u32 rows_cap = (u32)std::min<size_t>(e->group_cap, 0xFFFFFFFFu);
  {
    // ---- order of the batch's descriptors by home key (label code + truncated cell):
    // stable 8-bit radix passes over 12 + 3*cbits key bits, then one GroupRow of bucket
    // lookups per distinct home cell
    CHK(ensure(e, e->keyA, (size_t)n_slots * sizeof(u64)));
    CHK(ensure(e, e->keyB, (size_t)n_slots * sizeof(u64)));
    CHK(ensure(e, e->valA, (size_t)n_slots * sizeof(u32)));
    CHK(ensure(e, e->valB, (size_t)n_slots * sizeof(u32)));
    CHK(ensure(e, e->gid, (size_t)n_slots * sizeof(u32)));
    CHK(ensure(e, e->n_valid, sizeof(u32)));
    u64 *kin = e->keyA.as<u64>(), *kout = e->keyB.as<u64>();
    u32 *vin = e->valA.as<u32>(), *vout = e->valB.as<u32>();
    CHK(ensure(e, e->q_prefix, (size_t)nq * sizeof(u32)));
    CHK(ensure(e, e->group_first, (size_t)n_slots * sizeof(u32)));
    CHK(ensure(e, e->n_groups, sizeof(u32)));
    // pass slots: at most one per group and one per two descriptors (probe_kernels.hip.h)
    const bool pair = !e->diag && SGTD_PAIR >= 2;
    const size_t max_pass_slots = (size_t)pass_slot_count((u32)n_slots, (u32)n_slots, pair) + 64;
    CHK(ensure(e, e->pos_of_slot, max_pass_slots * sizeof(u32)));
    CHK(ensure(e, e->rec_off, max_pass_slots * sizeof(u32)));
    // 160 B per descriptor slot (110 used at the 10 000-frame default), 256 B for tables beyond 1e8 entries (more of
    // the 27 cells of a home cell have a bucket: 190 B used at 100 000 frames); grown on overflow
    if (e->pool_units == 0) e->pool_units = std::max<size_t>(65536, (size_t)n_slots * (e->n_entries > 100000000 ? 16 : 10));
    CHK(ensure(e, e->pass_pool, (e->pool_units + SGTD_PASS_SLACK_UNITS) * sizeof(uint4)));
    const u32 *nv = e->n_valid.as<u32>();
    if (small) {
      SmallOrder SO;
      SO.ctr = e->cursors.as<u32>(); SO.ctr_words = (u32)kCtrWords;
      SO.q_M = e->q_M.as<u32>(); SO.q_P = e->q_P.as<unsigned long long>();
      SO.votes = votes_per_query ? nullptr : e->votes.as<u32>(); SO.slot_of_words = e->slot_of.as<u32>(); SO.span = span;
      SO.cand_frame = e->cand_frame.as<int>(); SO.cand_votes = e->cand_votes.as<int>(); SO.cand_num = cn;
      SO.q_prefix = e->q_prefix.as<u32>(); SO.n_valid = e->n_valid.as<u32>(); SO.order = vin; SO.gid = e->gid.as<u32>();
      SO.group_first = e->group_first.as<u32>(); SO.n_groups = e->n_groups.as<u32>(); SO.pos_of_slot = e->pos_of_slot.as<u32>();
      SO.n_slots = (u32)n_slots; SO.max_pass_slots = (u32)max_pass_slots; SO.cbits = cbits; SO.sub_bits = sub_bits; SO.pair = pair ? 1 : 0; SO.key_bits = key_bits;
      SO.qrec = nullptr; SO.n_qrec = 0; SO.rough = e->dc.rough;
      const size_t lds = (size_t)SGTD_SMALL_SLOTS * 16;
      static const bool lds_set = [&] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&small_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; }();
      if (!lds_set) return SGTD_ERR_HIP;
      small_order_kernel<<<1, SGTD_SMALL_THREADS, lds, e->stream>>>(v.Q, SO);
      HIPCHK(hipGetLastError());
    } else {
    query_prefix_kernel<<<1, 256, 0, e->stream>>>(e->q_count.as<u32>(), e->q_prefix.as<u32>(), nq, e->n_valid.as<u32>());
    HIPCHK(hipGetLastError());
    if (key_bits <= 32) {      // 32-bit keys: a third less traffic in every sort pass
      u32 *k32 = reinterpret_cast<u32 *>(kin), *k32o = reinterpret_cast<u32 *>(kout);
      home_keys_kernel<u32><<<grid_for(n_slots, 256), 256, 0, e->stream>>>(v.Q, e->q_prefix.as<u32>(), k32, vin, n_slots, cbits, sub_bits);
      HIPCHK(hipGetLastError());
      CHK(radix_sort_pairs(e, k32, k32o, vin, vout, n_slots, key_bits, false, nv));   // only the n_valid compact elements are live
      group_heads_kernel<u32><<<grid_for(n_slots, 256), 256, 0, e->stream>>>(k32, nv, e->gid.as<u32>(), n_slots, cbits, sub_bits);
    } else {
      home_keys_kernel<u64><<<grid_for(n_slots, 256), 256, 0, e->stream>>>(v.Q, e->q_prefix.as<u32>(), kin, vin, n_slots, cbits, sub_bits);
      HIPCHK(hipGetLastError());
      CHK(radix_sort_pairs(e, kin, kout, vin, vout, n_slots, key_bits, false, nv));
      group_heads_kernel<u64><<<grid_for(n_slots, 256), 256, 0, e->stream>>>(kin, nv, e->gid.as<u32>(), n_slots, cbits, sub_bits);
    }
    HIPCHK(hipGetLastError());
    CHK(device_scan(e, e->gid.as<u32>(), e->gid.as<u32>(), n_slots));
    group_first_kernel<<<grid_for(n_slots, 256), 256, 0, e->stream>>>(e->gid.as<u32>(), nv, e->group_first.as<u32>(),
                                                                        e->n_groups.as<u32>(), n_slots);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(e->pos_of_slot.p, 0xFF, max_pass_slots * sizeof(u32), e->stream));
    pass_slots_kernel<<<grid_for(n_slots, 256), 256, 0, e->stream>>>(e->gid.as<u32>(), e->group_first.as<u32>(), nv,
                                                                       e->pos_of_slot.as<u32>(), n_slots, pair ? 1 : 0);
    HIPCHK(hipGetLastError());
    }
    e->thr2_pending = 0;
    // pass slots per wave ticket: about 1.5k entry visits (neighbouring home cells then go to different waves of one XCD at
    // about the same time and find each other's buckets in its L2: at six waves per SIMD tickets of 4 / 3 / 2 pass slots
    // fetch 8.0 / 6.0 / 4.2 GB per sweep of the default batch in the same 4.8-5.0 ms; 1: 3.1 GB in 6.0 ms), from the visits per descriptor the
    // previous batch measured (2 until there is one); SGTD_SORTED_CHUNK overrides
    u32 chunk = 2;
    if (e->stats.last_D > 0 && e->stats.last_P_swept > 0) {
      // last_P_swept counts a pass's shared list once: visits per pass ~ P_swept / (D / descriptors per pass)
      const double per_pass = (double)e->stats.last_P_swept / ((double)e->stats.last_D / (pair ? (SGTD_PAIR >= 4 ? 3.2 : 1.9) : 1.0));
      chunk = (u32)std::min(8.0, std::max(1.0, std::floor(1536.0 / per_pass + 0.5)));
    }
    if (e->sorted_chunk > 0) chunk = (u32)std::min(SGTD_TICKET_MAX, e->sorted_chunk);
    // the grid is sized by resident waves, not by work items: every wave pulls tickets
    int sgrid = e->n_cus * 8;
    if (const char *o = getenv("SGTD_SWEEP_BLOCKS_PER_CU")) sgrid = e->n_cus * std::max(1, atoi(o));   // experiment knob
    PassPool PP;
    PP.pool = e->pass_pool.as<uint4>(); PP.rec_off = e->rec_off.as<u32>();
    PP.cursor = v.B.pool_cursor(); PP.cap = (u32)std::min<size_t>(e->pool_units, 0xFFFFFF00u);
    {
      // one GroupRow per home cell (both segments' directory rows), the passes' visit lists from it, then
      // ONE sweep: a pass's list runs through the main segment's ranges and then the tail's, cell by cell
      Views &vs = v;
      group_resolve_kernel<<<e->n_cus * 16, 256, 0, e->stream>>>(vs.T, vs.Q, vin, e->group_first.as<u32>(),
                                                                  e->n_groups.as<u32>(), nv, e->cell_rows.as<unsigned char>(), rows_cap, vs.B.overflow());
      HIPCHK(hipGetLastError());
      // resident workgroups (five waves per SIMD: 100 vector registers, 7 KB of staged rows per wave) x 4 rounds,
      // grid-stride over the slots
      int plan_per_cu = 20;
      if (const char *o = getenv("SGTD_PLAN_BLOCKS_PER_CU")) plan_per_cu = std::max(1, atoi(o));   // experiment knob
      const int pgrid = (int)std::min<long long>(grid_for((long long)max_pass_slots, SGTD_PLAN_THREADS), (long long)e->n_cus * plan_per_cu);
#define SGTD_LAUNCH_PLAN(PR, TL)                                                                               \
  plan_passes_kernel<PR, TL><<<pgrid, SGTD_PLAN_THREADS, 0, e->stream>>>(vs.T, vs.Q, vin, e->gid.as<u32>(), e->pos_of_slot.as<u32>(), nv, \
                                                       e->n_groups.as<u32>(), rows, rows_cap, PP, vs.B.n_visit, vs.B.list,          \
                                                       vs.B.overflow())
      if (vs.T.tail_off) { if (pair) SGTD_LAUNCH_PLAN(true, true); else SGTD_LAUNCH_PLAN(false, true); }
      else { if (pair) SGTD_LAUNCH_PLAN(true, false); else SGTD_LAUNCH_PLAN(false, false); }
#undef SGTD_LAUNCH_PLAN
      HIPCHK(hipGetLastError());
      if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_SORT], e->stream));   // ms_probe = the sweep from here
      // 32-bit byte offsets into the probe layout while it stays below 4 GB (record addresses
      // are a wave-uniform base + a 32-bit lane offset either way)
      const bool narrow = ((unsigned long long)vs.T.n_entries + SGTD_SENTINELS + (vs.T.tail_off ? vs.T.n_entries1 + SGTD_SENTINELS : 0u)) * sizeof(HotEntry) < (1ull << 32);
      // can a descriptor of the batch carry a frame id the table holds?  Frames built by
      // sgtd_query_frames are stamped with the current frame id (one beyond the newest map frame
      // in the reference's use); descriptors handed in by the caller carry whatever they carry
      const bool frames = e->last_kind != 1 || (e->have_frames && e->last_qframe >= e->frame_lo && e->last_qframe <= e->frame_hi);
#define SGTD_LAUNCH_SORTED(DG, WD, FR)                                                                          \
  probe_sorted_kernel<DG, WD, FR><<<sgrid, SGTD_PROBE_THREADS, 0, e->stream>>>(                                 \
      vs.T, vs.B, vs.Q, PP, e->dc.rough, e->n_valid.as<u32>(), e->n_groups.as<u32>(), chunk)
      if (e->diag) SGTD_LAUNCH_SORTED(true, true, true);
      else if (narrow && frames) SGTD_LAUNCH_SORTED(false, false, true);
      else if (narrow) SGTD_LAUNCH_SORTED(false, false, false);
      else if (frames) SGTD_LAUNCH_SORTED(false, true, true);
      else SGTD_LAUNCH_SORTED(false, true, false);
#undef SGTD_LAUNCH_SORTED
      HIPCHK(hipGetLastError());
      resolve_undecided_kernel<<<64, 256, 0, e->stream>>>(vs.T, vs.Q, vs.B, e->q_M.as<u32>());
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipGetLastError());
#ifdef SGTD_EXP_PHASE
    {
      static int pcalls = 0;
      if (++pcalls == 6) {
        HIPCHK(hipStreamSynchronize(e->stream));
        unsigned long long ph[8];
        HIPCHK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph)));
        const double tot = (double)ph[7];
        fprintf(stderr, "PHASE (fractions of wave life, %llu waves): prologue %.3f locate+issue %.3f wait-loads %.3f test %.3f epilogue %.3f between-passes %.3f rest %.3f\n",
                ph[6], ph[0] / tot, ph[1] / tot, ph[2] / tot, ph[3] / tot, ph[4] / tot, ph[5] / tot,
                (tot - ph[0] - ph[1] - ph[2] - ph[3] - ph[4] - ph[5]) / tot);
      }
    }
#endif
    if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_PROBE], e->stream));
    // one 16-wave workgroup per (query, frame tile) when that fills the chip: the tile's LDS
    // histogram is final (no flush atomics); spans beyond LDS take several tiles of 36 Ki frames
    if (fused_votes) {
      const size_t lds = votes_topk_lds_bytes(span);
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&votes_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      votes_topk_kernel<<<nq, SGTD_VT_THREADS, lds, e->stream>>>(v.Q, v.B, span, v.T.frame_lo, blocks, cn, e->q_M.as<u32>(),
                                                                  e->q_P.as<unsigned long long>(), e->n_cand.as<int>(), e->cand_frame.as<int>(),
                                                                  e->cand_votes.as<int>(), e->pair_off.as<long long>(), e->q_pairs.as<u32>());
    } else if (votes_per_query) {
      const size_t tile_bytes = (size_t)tile_span * sizeof(u32);
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&votes_query_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));
      votes_query_kernel<<<nq * (int)n_tiles, SGTD_VOTES_Q_THREADS, tile_bytes, e->stream>>>(
          v.Q, v.B, span, v.T.frame_lo, tile_span, blocks, e->q_M.as<u32>(), e->q_P.as<unsigned long long>());
    } else if (lds_votes) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&votes_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_bytes));
      votes_kernel<true><<<agrid, 256, hist_bytes, e->stream>>>(v.Q, v.B, span, v.T.frame_lo, blocks, e->q_M.as<u32>(),
                                                                e->q_P.as<unsigned long long>());
    } else {
      votes_kernel<false><<<agrid, 256, 0, e->stream>>>(v.Q, v.B, span, v.T.frame_lo, blocks, e->q_M.as<u32>(),
                                                        e->q_P.as<unsigned long long>());
    }
    HIPCHK(hipGetLastError());
    if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_VOTES], e->stream));
  }
  if (!fused_votes) {
    topk_kernel<<<nq, SGTD_TOPK_THREADS, 0, e->stream>>>(e->votes.as<u32>(), span, v.T.frame_lo, cn, e->n_cand.as<int>(),
                                            e->cand_frame.as<int>(), e->cand_votes.as<int>(),
                                            e->slot_of.as<unsigned char>());
    HIPCHK(hipGetLastError());
  }
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_TOPK], e->stream));
  CHK(export_candidates(e, v));      // (multi-GPU step: the local tables start travelling before the lists are written)
  e->lists_pending = false;
  e->lists_masked = false;
  e->fused_votes_last = fused_votes;
  if (fused_pairs) {
    // the lists' offsets are the prefix sums of the candidates' votes; the lists themselves by one workgroup per query
    e->pairs_per_query = true;
    e->stats.select_form = fused_votes ? 2 : 1;
    e->batch_valid = true;
    if (e->defer_lists) {
      // the caller writes the lists itself once it knows which candidates survive the merge (sgtd_finish_lists)
      if (e->timing)
        for (int k : {EV_COUNT_T, EV_SCAN, EV_WRITE}) HIPCHK(hipEventRecord(e->ev[k], e->stream));
      batch_totals_kernel<<<1, 1, 0, e->stream>>>(v.B.ctr, e->totals.as<unsigned long long>());
      HIPCHK(hipGetLastError());
      e->lists_pending = true;
      e->verified = false;
      e->batch_synced = false;
      return SGTD_OK;
    }
    return launch_lists(e, v, nullptr, /*first=*/true);
  }
  e->pairs_per_query = false;
  e->stats.select_form = 0;
  CompactLists CL;
  CL.pair = e->c_pair.as<u64>();
  CL.blk_start = e->c_blk.as<u32>(); CL.blk_n = e->c_blk.as<u32>() + (size_t)nq * blocks;
  CL.cursor = v.B.compact_cursor(); CL.cap = (u32)std::min<size_t>(e->c_pair.bytes / sizeof(u64), 0xFFFFFFF0u);
  // the compact lists' words: 4 bytes where an entry's rank among its frame's fits 19 bits, else 8
  const bool narrow_pairs = v.T.map.bits <= SGTD_NARROW_RANK_BITS && !e->wide_pairs;
  if (span <= 48 * 1024) {
    const int sl_bytes = (int)((span + 15) & ~15u);
#define SGTD_LAUNCH_COUNT(NP)                                                                                          \
  do {                                                                                                                 \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&block_count_kernel<true, NP>),                          \
                               hipFuncAttributeMaxDynamicSharedMemorySize, sl_bytes));                                 \
    block_count_kernel<true, NP><<<agrid, 256, sl_bytes, e->stream>>>(v.Q, v.B, e->n_cand.as<int>(), e->cand_frame.as<int>(), cn, \
                                                                      blocks, e->blk_count.as<u32>(), CL, nullptr, nullptr,      \
                                                                      e->slot_of.as<unsigned char>(), span, v.T.frame_lo);        \
  } while (0)
    if (narrow_pairs) SGTD_LAUNCH_COUNT(true); else SGTD_LAUNCH_COUNT(false);
#undef SGTD_LAUNCH_COUNT
  } else {
    if (narrow_pairs)
      block_count_kernel<false, true><<<agrid, 256, 0, e->stream>>>(v.Q, v.B, e->n_cand.as<int>(), e->cand_frame.as<int>(), cn,
                                                                    blocks, e->blk_count.as<u32>(), CL, nullptr, nullptr,
                                                                    e->slot_of.as<unsigned char>(), span, v.T.frame_lo);
    else
      block_count_kernel<false, false><<<agrid, 256, 0, e->stream>>>(v.Q, v.B, e->n_cand.as<int>(), e->cand_frame.as<int>(), cn,
                                                                     blocks, e->blk_count.as<u32>(), CL, nullptr, nullptr,
                                                                     e->slot_of.as<unsigned char>(), span, v.T.frame_lo);
  }
  HIPCHK(hipGetLastError());
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_COUNT_T], e->stream));
  block_scan_kernel<<<nq, 64, 0, e->stream>>>(e->blk_count.as<u32>(), blocks, cn, e->n_cand.as<int>(),
                                               e->pair_off.as<long long>(), e->q_pairs.as<u32>(),
                                               v.B.overflow(), nq == 1 ? e->q_pair_base.as<u32>() : nullptr, (u32)std::min<size_t>(e->pair_cap, 0xFFFFFFF0u),
                                               nq == 1 ? e->totals.as<unsigned long long>() : nullptr);
  HIPCHK(hipGetLastError());
  if (nq != 1) {
    query_base_kernel<<<1, 256, 0, e->stream>>>(e->q_pairs.as<u32>(), e->q_pair_base.as<u32>(), nq,
                                                 (u32)std::min<size_t>(e->pair_cap, 0xFFFFFFF0u), v.B.overflow());
    HIPCHK(hipGetLastError());
  }
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_SCAN], e->stream));
  CHK(launch_block_write(e, v, CL, agrid, blocks));
  HIPCHK(hipGetLastError());
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_WRITE], e->stream));
  if (nq != 1) {
    batch_totals_kernel<<<1, 1, 0, e->stream>>>(v.B.ctr, e->totals.as<unsigned long long>());
    HIPCHK(hipGetLastError());
  }
  e->batch_valid = true;
  e->verified = false;
  e->batch_synced = false;
  return SGTD_OK;
}

int enqueue_frames(sgtd_engine *e) {
  // (re)runs the last frames batch: build on device, then select
  const int nq = e->nq;
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_START], e->stream));
  CHK(launch_build(e, e->last_xyz, e->last_label, e->kp_off_dev.as<long long>(), nq, e->last_max_n,
                   e->last_qframe, 0, e->qd.view(), e->q_stride, e->q_count.as<u32>()));
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_BUILD], e->stream));
  return launch_select(e);
}

int rerun(sgtd_engine *e) {
  if (e->last_kind == 1) return enqueue_frames(e);
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_START], e->stream));
  return launch_select(e);
}

// the candidate-pair buffer was too small: everything up to the per-block counts is intact,
// only the output offsets and the write pass run again
int rerun_write(sgtd_engine *e) {
  const int nq = e->nq;
  const int blocks = (int)((e->q_stride + block_chunk(e) - 1) / block_chunk(e));
  const int groups = (blocks + 3) / 4;
  const int agrid = ((nq + 7) / 8) * groups * 8;
  CHK(ensure(e, e->pairs, e->pair_cap * sizeof(u64)));
  Views v = make_views(e);
  HIPCHK(hipMemsetAsync(v.B.overflow() + 1, 0, sizeof(int), e->stream));
  if (e->pairs_per_query) return launch_lists(e, v, e->lists_masked ? e->list_keep : nullptr, /*first=*/false);
  CompactLists CL;
  CL.pair = e->c_pair.as<u64>();
  CL.blk_start = e->c_blk.as<u32>(); CL.blk_n = e->c_blk.as<u32>() + (size_t)nq * blocks;
  CL.cursor = v.B.compact_cursor(); CL.cap = (u32)std::min<size_t>(e->c_pair.bytes / sizeof(u64), 0xFFFFFFF0u);
  query_base_kernel<<<1, 256, 0, e->stream>>>(e->q_pairs.as<u32>(), e->q_pair_base.as<u32>(), nq,
                                               (u32)std::min<size_t>(e->pair_cap, 0xFFFFFFF0u), v.B.overflow());
  HIPCHK(hipGetLastError());
  CHK(launch_block_write(e, v, CL, agrid, blocks));
  return SGTD_OK;
}

// A handle attached to another one's table (sgtd_attach_table) holds that table's device pointers.  Whatever changes the
// owner's table — an add, a load, a finalize that rebuilds or merges a segment (its own fifth batch on a tail does) — may
// have freed them: every call of the view that would touch the table again (a batch's re-run, the verification, gathers of
// entries) asks here first.  (Work the view had already enqueued is safe: hipFree waits for the device.)
int view_current(sgtd_engine *e) {
  if (!e->attached_to) return SGTD_OK;
  if (e->attached_to->table_version != e->attached_version || !e->attached_to->finalized) {
    e->err = "the owner's table changed since sgtd_attach_table: attach again";
    e->batch_valid = false;
    return SGTD_ERR_STATE;
  }
  return SGTD_OK;
}

// the batch's stage times from its events (the batch has been waited for)
void stage_times(sgtd_engine *e) {
  if (!e->timing) return;
  sgtd_stats &s = e->stats;
  auto el = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, e->ev[a], e->ev[b]); return ms; };
  if (e->last_kind == 1) { s.ms_build = el(EV_START, EV_BUILD); s.ms_sort = el(EV_BUILD, EV_SORT); }
  else { s.ms_build = 0; s.ms_sort = el(EV_START, EV_SORT); }
  s.ms_probe = el(EV_SORT, EV_PROBE); s.ms_votes = el(EV_PROBE, EV_VOTES);
  s.ms_topk = el(EV_VOTES, EV_TOPK); s.ms_count = el(EV_TOPK, EV_COUNT_T);
  s.ms_scan = el(EV_COUNT_T, EV_SCAN); s.ms_write = el(EV_SCAN, EV_WRITE);
  s.ms_total = el(EV_START, EV_WRITE);
}

int sync_batch(sgtd_engine *e) {
  PinScope pin_scope(e);
  if (!e->batch_valid) return SGTD_ERR_STATE;
  CHK(view_current(e));
  if (e->batch_synced) return SGTD_OK;
  e->stats.overflowed = 0;
  unsigned long long swept = 0;
  for (int attempt = 0; attempt < 12; attempt++) {
    int ovf[2] = {0, 0};
    unsigned long long cursor = 0, need = 0;
    swept = 0;
    u32 total = 0, pool_used = 0;
    u32 ctr[12];     // ProbeBuffers::ctr, one copy
    u32 n_groups = 0;
    unsigned long long tot[3] = {0, 0, 0};
    // (deferred lists that nobody finished, or a re-run that changed the candidates: the lists of all of them)
    if (e->lists_pending) CHK(launch_lists(e, make_views(e), nullptr, /*first=*/false));
    CHK(d2h(e, ctr, e->cursors.p, sizeof(ctr)));
    if (e->totals.p) CHK(d2h(e, tot, e->totals.p, sizeof(tot)));
    if (e->n_groups.p) CHK(d2h(e, &n_groups, e->n_groups.p, sizeof(u32)));
    CHK(d2h(e, &total, e->q_pair_base.as<u32>() + e->nq, sizeof(u32)));
    CHK(xfer_sync(e));
    std::memcpy(&cursor, ctr, 8); std::memcpy(&need, ctr + 4, 8); std::memcpy(&swept, ctr + 6, 8);
    cursor <<= SGTD_REC_SHIFT;          // (the device counts granules of four records)
    pool_used = ctr[8]; ovf[0] = (int)ctr[10]; ovf[1] = (int)ctr[11];
    e->stats.last_list_moves = ctr[9];
    e->stats.batches_total = (int64_t)tot[0]; e->stats.overflow_launches_total = (int64_t)tot[1]; e->stats.list_moves_total = (int64_t)tot[2];
    if (!ovf[0] && !ovf[1]) {
      // slab use varies a little from run to run (which wave sweeps what): when a batch comes within
      // a tenth of the capacity, make room for half as much again (reallocated at the next launch)
      const size_t lim0 = kRecLimit;
      if ((double)cursor * 1.1 > (double)e->rec_cap) e->rec_cap = std::min<size_t>(lim0, (size_t)((double)cursor * 1.5));
      break;
    }
    e->stats.overflowed = 1;
    if (ovf[0]) e->stats.reruns_total++; else e->stats.rewrites_total++;
    if (getenv("SGTD_DEBUG"))
      fprintf(stderr, "sgtd: batch re-run (attempt %d): flags %d %d, records %llu of %zu (+%llu wanted), pass pool %u of %zu units, home cells %u of %zu rows, pairs %u of %zu\n",
              attempt, ovf[0], ovf[1], cursor, e->rec_cap, need, pool_used, e->pool_units, n_groups, e->group_cap, total, e->pair_cap);
    if (attempt == 11) return SGTD_ERR_CAPACITY;
    // grow towards the index limits; a batch that does not fit even there must be split
    const size_t lim = kRecLimit, pair_lim = kIndexLimit;
    if (ovf[0] && (size_t)n_groups > e->group_cap) {
      // more distinct home cells than GroupRows were reserved
      e->group_cap = (size_t)n_groups + (size_t)n_groups / 4 + 1024;
    } else if (ovf[0] && (size_t)pool_used > e->pool_units) {
      // the pass records did not fit (the cursor kept counting: pool_used is what the batch needs)
      if (e->pool_units >= 0xFFFFFF00ull) return SGTD_ERR_CAPACITY;
      e->pool_units = std::min<size_t>(0xFFFFFF00ull, (size_t)pool_used + (size_t)pool_used / 4 + 65536);
    } else if (ovf[0]) {
      // The cursor counts the room the lists RESERVED (rec_rate / 256 of every visit list), `need` the matches that
      // found none.  Reservations far beyond the buffer with few matches left over: the rate is too generous for
      // this table (a first batch on long buckets that match little: skewed label frequencies) — a list that
      // outgrows its room moves, so the rate can drop safely; only when it is at its floor is the batch too large.
      const bool reservations = (double)cursor > 1.5 * (double)e->rec_cap + 4.0 * (double)need;
      if (reservations && e->rec_rate_cap > 2) {
        e->rec_rate_cap = std::max<u32>(2, std::min<u32>(e->rec_rate_cap, 64) / 4);
      } else if (e->rec_cap >= lim) {
        return SGTD_ERR_CAPACITY;
      }
      // what was stored fits rec_cap, `need` matches did not; slabs leave about an eighth unused,
      // every wave strands part of its last slab
      const size_t want = (size_t)((double)(e->rec_cap + need) * 1.4) + (size_t)e->n_cus * 32 * SGTD_PAIR * 512;
      e->rec_cap = std::min<size_t>(lim, std::max<size_t>(reservations ? e->rec_cap : e->rec_cap * 2, want));
    } else if (ovf[1]) {
      if (e->pair_cap >= pair_lim) return SGTD_ERR_CAPACITY;
      e->pair_cap = std::min<size_t>(pair_lim, std::max<size_t>(e->pair_cap * 2, (size_t)total + (total >> 3) + 65536));
      CHK(rerun_write(e));
      continue;
    }
    CHK(rerun(e));
  }
  const int nq = e->nq, cn = e->dc.cand_num;
  if (!(e->h_count.resize(nq) && e->h_pair_base.resize(nq + 1) && e->h_q_M.resize(nq) && e->h_q_P.resize(nq) &&
        e->h_n_cand.resize(nq) && e->h_cand_frame.resize((size_t)nq * cn) && e->h_cand_votes.resize((size_t)nq * cn) &&
        e->h_pair_off.resize((size_t)nq * (cn + 1)))) {
    e->err = "hipHostMalloc of the result tables failed";
    return SGTD_ERR_HIP;
  }
  // the batch's result tables: eight direct DMA transfers into page-locked arrays the handle keeps
  auto pull = [&](void *dst, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream) : hipSuccess; };
  HIPCHK(pull(e->h_count.data(), e->q_count.p, nq * sizeof(u32)));
  HIPCHK(pull(e->h_pair_base.data(), e->q_pair_base.p, (nq + 1) * sizeof(u32)));
  HIPCHK(pull(e->h_q_M.data(), e->q_M.p, nq * sizeof(u32)));
  HIPCHK(pull(e->h_q_P.data(), e->q_P.p, nq * sizeof(unsigned long long)));
  HIPCHK(pull(e->h_n_cand.data(), e->n_cand.p, nq * sizeof(int)));
  HIPCHK(pull(e->h_cand_frame.data(), e->cand_frame.p, (size_t)nq * cn * sizeof(int)));
  HIPCHK(pull(e->h_cand_votes.data(), e->cand_votes.p, (size_t)nq * cn * sizeof(int)));
  HIPCHK(pull(e->h_pair_off.data(), e->pair_off.p, (size_t)nq * (cn + 1) * sizeof(long long)));
  CHK(xfer_sync(e));
  sgtd_stats &s = e->stats;
  s.last_queries = nq;
  s.last_D = 0; s.last_P = 0; s.last_M = 0; s.last_cand_pairs = 0;
  s.last_P_swept = (int64_t)swept;
  for (int q = 0; q < nq; q++) {
    s.last_D += e->h_count[q];
    s.last_P += (int64_t)e->h_q_P[q];
    s.last_M += e->h_q_M[q];
    s.last_cand_pairs += e->h_pair_off[(size_t)q * (cn + 1) + cn];
  }
  stage_times(e);
  e->batch_synced = true;
  return SGTD_OK;
}

// A pending query batch keeps its inputs in the handle's staging buffers for a re-run after a
// work-buffer overflow: anything that is about to reuse them settles the batch first.
// Expected rough matches of one query frame of n_keypoints keypoints, before any batch has been
// measured.  A query descriptor distributed like the table's entries meets buckets of
// L = sum(len^2) / E entries on average; its matches are the entries inside its threshold ball:
// about 0.55 L for the shipped rough_dis_threshold 0.03 (ball of ~0.5 m in cells of 1 m^3; the
// ball's volume grows with the cube of the threshold) plus the re-observations of the same
// place, ~65 per descriptor on the synthetic maps (measured 73 / 166 matches per descriptor at
// 1 k / 10 k frames: the estimate gives 77 / 185).  About 62 % of the 36 n_keypoints triplets
// survive the length filter and the dedup.
double est_matches_per_query(const sgtd_engine *e, int n_keypoints) {
  if (e->stats.last_queries > 0 && e->stats.last_D > 0 && e->nq > 0 && e->q_stride > 0) {
    // measured: matches per descriptor slot of the last batch, scaled to this frame size
    const double per_slot = (double)e->stats.last_M / ((double)e->stats.last_queries * (double)e->q_stride);
    return per_slot * (double)n_keypoints * e->dc.tpi;
  }
  if (e->n_entries <= 0) return 0.0;
  const double L = (e->seg[0].sum_len_sq + (e->n_seg > 1 ? e->seg[1].sum_len_sq : 0.0)) / (double)e->n_entries;
  const double r = e->dc.rough / 0.03;
  const double per_desc = 0.55 * L * std::min(27.0, r * r * r) + 65.0;
  return 0.62 * (double)n_keypoints * e->dc.tpi * per_desc;
}

int settle_pending(sgtd_engine *e) {
  if (e->batch_valid && !e->batch_synced) return sync_batch(e);
  return SGTD_OK;
}

// A tail segment makes every visit list run through two sets of buckets (+11 % step time).  While appends
// keep arriving that is the cheap side of the trade (each append sorts only the tail); once
// SGTD_TAIL_BATCHES batches in a row have used the same tail the next one merges it into the main segment
// (one full build, ~0.25 ms per million entries).
#define SGTD_TAIL_BATCHES 4
int settle_tail(sgtd_engine *e) {
  if (e->attached_to) return view_current(e);
  if (e->finalized && e->n_seg == 2) {
    if (e->tail_batches >= SGTD_TAIL_BATCHES) { e->tail_batches = 0; return do_finalize(e, /*force_merge=*/true); }
    e->tail_batches++;
  } else {
    e->tail_batches = 1;    // no tail, or an append since the last batch (do_finalize rebuilds the tail): its first batch
  }
  return do_finalize(e);
}

int check_cfg(const sgtd_config *c) {
  if (c->descriptor_near_num < 3 || c->descriptor_near_num > SGTD_MAX_K) return SGTD_ERR_UNSUPPORTED;
  if (c->candidate_num < 1 || c->candidate_num > SGTD_MAX_CAND) return SGTD_ERR_UNSUPPORTED;
  if (c->max_frame_n < 1) return SGTD_ERR_INVALID;
  if (!(c->std_side_resolution > 0) || !(c->descriptor_max_len > 0)) return SGTD_ERR_INVALID;
  if (!(c->descriptor_max_len * 1000.0 < 2097151.0)) return SGTD_ERR_UNSUPPORTED;
  if (!(c->descriptor_max_len / c->std_side_resolution + 2.0 < 65535.0)) return SGTD_ERR_UNSUPPORTED;
  return SGTD_OK;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
namespace {
// one SoA field <-> file, staged through a bounded host buffer
template <class T>
int stream_field(sgtd_engine *e, FILE *f, T *dev, size_t n_items, bool to_file, std::vector<char> &stage) {
  const size_t chunk = stage.size() / sizeof(T);
  for (size_t o = 0; o < n_items; o += chunk) {
    const size_t m = std::min(chunk, n_items - o);
    if (to_file) {
      HIPCHK(hipMemcpyAsync(stage.data(), dev + o, m * sizeof(T), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      if (fwrite(stage.data(), sizeof(T), m, f) != m) return SGTD_ERR_IO;
    } else {
      if (fread(stage.data(), sizeof(T), m, f) != m) return SGTD_ERR_IO;
      HIPCHK(hipMemcpyAsync(dev + o, stage.data(), m * sizeof(T), hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
    }
  }
  return SGTD_OK;
}
int stream_table(sgtd_engine *e, FILE *f, size_t n, bool to_file) {
  std::vector<char> stage((size_t)64 << 20);
  CHK(stream_field(e, f, e->tab.side.as<double>(), n * 3, to_file, stage));
  CHK(stream_field(e, f, e->tab.angle.as<double>(), n * 3, to_file, stage));
  CHK(stream_field(e, f, e->tab.center.as<double>(), n * 3, to_file, stage));
  CHK(stream_field(e, f, e->tab.vertex.as<float>(), n * 9, to_file, stage));
  CHK(stream_field(e, f, e->tab.label.as<int>(), n * 3, to_file, stage));
  CHK(stream_field(e, f, e->tab.frame.as<u32>(), n, to_file, stage));
  CHK(stream_field(e, f, e->tab.node_id.as<int>(), n * 3, to_file, stage));
  return SGTD_OK;
}
}  // namespace

extern "C" {

void sgtd_default_config(sgtd_config *cfg) {
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->descriptor_near_num = 10;
  cfg->candidate_num = 50;
  cfg->max_frame_n = 20000;
  cfg->device_id = 0;
  cfg->descriptor_min_len = 0.5;
  cfg->descriptor_max_len = 50.0;
  cfg->std_side_resolution = 1.0;
  cfg->rough_dis_threshold = 0.03;
  cfg->first_frame_id = 0;
}

const char *sgtd_strerror(int status) {
  switch (status) {
    case SGTD_OK: return "ok";
    case SGTD_ERR_INVALID: return "invalid argument";
    case SGTD_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
    case SGTD_ERR_HIP: return "HIP runtime error";
    case SGTD_ERR_CAPACITY: return "buffer capacity exceeded";
    case SGTD_ERR_FRAME_LIMIT: return "frame id beyond max_frame_n";
    case SGTD_ERR_UNSUPPORTED: return "configuration outside the kernels' envelope";
    case SGTD_ERR_STATE: return "call order";
    case SGTD_ERR_IO: return "graph file could not be opened or parsed";
    default: return "unknown status";
  }
}

const char *sgtd_last_error(sgtd_handle h) { return h ? h->err.c_str() : ""; }

uint32_t sgtd_label_code(int a, int b, int c) { return label_code(a, b, c); }
uint64_t sgtd_table_key(uint32_t code, uint32_t x, uint32_t y, uint32_t z) { return pack_key(code, x, y, z); }
uint64_t sgtd_dedup_key(uint64_t mx, uint64_t my, uint64_t mz) { return pack_milli_key(mx, my, mz); }

int sgtd_create(const sgtd_config *cfg, sgtd_handle *out) {
  if (!cfg || !out) return SGTD_ERR_INVALID;
  *out = nullptr;
  int r = check_cfg(cfg);
  if (r != SGTD_OK) return r;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SGTD_ERR_NO_DEVICE;
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return SGTD_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) return SGTD_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SGTD_ERR_NO_DEVICE;
  if (hipSetDevice(cfg->device_id) != hipSuccess) return SGTD_ERR_NO_DEVICE;
  sgtd_engine *e = new sgtd_engine();
  e->cfg = *cfg;
  e->dc.K = cfg->descriptor_near_num;
  e->dc.tpi = (cfg->descriptor_near_num - 1) * (cfg->descriptor_near_num - 2) / 2;
  e->dc.cand_num = cfg->candidate_num;
  e->dc.min_len = cfg->descriptor_min_len;
  e->dc.max_len = cfg->descriptor_max_len;
  e->dc.scale = 1.0 / cfg->std_side_resolution;  // STDesc.cpp:178
  e->dc.rough = cfg->rough_dis_threshold;
  e->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  e->current_frame_id = cfg->first_frame_id;
  e->qd.with_thr2 = true;
  if (const char *o = getenv("SGTD_SORTED_CHUNK")) e->sorted_chunk = atoi(o);
  if (const char *o = getenv("SGTD_TAIL_MAX")) e->tail_max = atoll(o);
  if (const char *o = getenv("SGTD_COARSE_AT")) e->coarse_at = (u32)std::min(62ll, std::max(0ll, atoll(o)));
  if (const char *o = getenv("SGTD_REC_RATE")) e->rec_rate_hook = (u32)std::min(256, std::max(1, atoi(o)));
  if (const char *o = getenv("SGTD_WIDE_PAIRS")) e->wide_pairs = atoi(o) != 0;
  if (const char *o = getenv("SGTD_SELECT_MODE")) e->select_mode = std::min(2, std::max(0, atoi(o)));
  if (const char *o = getenv("SGTD_WHOLE_AT")) e->whole_at = (u32)std::min(62ll, std::max(0ll, atoll(o)));
  // test hook: start with a small match-record buffer so that the overflow / re-run path runs
  if (const char *o = getenv("SGTD_REC_CAP")) { e->rec_cap = (size_t)std::max(1024ll, atoll(o)); e->rec_cap_fixed = true; }
  // test hooks: start with a tiny pass pool / GroupRow reservation so that their overflow / re-run paths run
  if (const char *o = getenv("SGTD_POOL_UNITS")) e->pool_units = (size_t)std::max(64ll, atoll(o));
  if (const char *o = getenv("SGTD_GROUP_CAP")) e->group_cap_hook = (size_t)std::max(1ll, atoll(o));
  if (const char *o = getenv("SGTD_PAIR_CAP")) { e->pair_cap = (size_t)std::max(64ll, atoll(o)); e->rec_cap_fixed = true; }
  for (int i = 0; i < EV_COUNT; i++)
    if (hipEventCreate(&e->ev[i]) != hipSuccess) { delete e; return SGTD_ERR_HIP; }
  if (hipEventCreateWithFlags(&e->ev_cand, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_export_free, hipEventDisableTiming) != hipSuccess) { delete e; return SGTD_ERR_HIP; }
  *out = e;
  return SGTD_OK;
}

int sgtd_create_multi(const sgtd_config *cfg, const int *device_ids, int n_dev, sgtd_handle *out) {
  if (n_dev == 1 && cfg && device_ids && out) {   // one device: the ordinary handle
    sgtd_config c = *cfg;
    c.device_id = device_ids[0];
    return sgtd_create(&c, out);
  }
  return multi::create(cfg, device_ids, n_dev, out);
}

int sgtd_device_count(sgtd_handle e) { return !e ? 0 : (e->grp ? multi::device_count(e) : 1); }

sgtd_handle sgtd_device_handle(sgtd_handle e, int k) {
  if (!e) return nullptr;
  if (!e->grp) return k == 0 ? e : nullptr;
  return multi::device_handle(e, k);
}

int sgtd_destroy(sgtd_handle e) {
  if (e && e->grp) return multi::destroy(e);
  if (!e) return SGTD_OK;
  if (e->n_views > 0) { e->err = "other handles are attached to this handle's table (sgtd_attach_table): destroy them first"; return SGTD_ERR_STATE; }
  (void)hipSetDevice(e->cfg.device_id);
  (void)hipStreamSynchronize(e->stream);
  if (e->attached_to) { e->attached_to->n_views--; e->attached_to = nullptr; }
  free_store(e->tab); free_store(e->tmp); free_store(e->qd); free_store(e->fetch); free_buf(e->fetch_idx);
  DevBuf *bufs[] = {&e->seg[0].hot, &e->seg[0].perm, &e->seg[0].hash, &e->seg[0].bucket_start, &e->seg[0].bucket_key, &e->seg[0].dir,
                    &e->seg[1].hot, &e->seg[1].perm, &e->seg[1].hash, &e->seg[1].bucket_start, &e->seg[1].bucket_key, &e->seg[1].dir, &e->slice_of, &e->sq_sum,
                    &e->keyA, &e->keyB, &e->valA, &e->valB, &e->hist, &e->digit_tot, &e->flags, &e->bad_flag,
                    &e->kp_off_dev, &e->xyz_dev, &e->label_dev, &e->b_kp_off_dev, &e->b_xyz_dev, &e->b_label_dev, &e->ws_keys, &e->ws_slots, &e->cnt_scan,
                    &e->tmp_count, &e->q_count, &e->n_valid, &e->cell_rows, &e->gid, &e->q_prefix, &e->group_first, &e->n_groups, &e->pos_of_slot, &e->rec_off, &e->pass_pool, &e->v_score, &e->v_pose, &e->v_inlier, &e->v_best, &e->inl_pairs, &e->inl_off, &e->v_hyp64, &e->v_hyp32, &e->v_bound, &e->v_hypB, &e->v_tau, &e->v_words, &e->v_okey[0], &e->v_okey[1], &e->v_oval[0], &e->v_oval[1], &e->cursors, &e->list, &e->n_visit,
                    &e->votes, &e->slot_of, &e->q_M, &e->q_P, &e->q_pairs, &e->q_pair_base,
                    &e->blk_count, &e->c_pair, &e->c_blk, &e->amb_queue, &e->rec, &e->rec_cell, &e->rec_dis, &e->rough_qi,
                    &e->rough_entry, &e->rough_frame, &e->rough_cell, &e->rough_dis, &e->n_cand, &e->cand_frame,
                    &e->cand_votes, &e->pair_off, &e->pairs, &e->totals, &e->inl_counts, &e->in_block, &e->b_in, &e->b_out,
                    // (the entry-id map: missing from this list until the engine's host code ran under the sanitizers — every destroyed
                    // handle kept them, 8 bytes per map frame and, with frame ids out of insertion order, 8 bytes per entry)
                    &e->frame_first, &e->by_frame, &e->id_of_g, &e->longest};
  for (DevBuf *b : bufs) free_buf(*b);
  for (auto &b : e->scan_lvl) free_buf(b);
  if (e->pin) (void)hipHostFree(e->pin);
  if (e->frame_host) (void)hipHostFree(e->frame_host);
  if (e->pin_build) (void)hipHostFree(e->pin_build);
  for (int i = 0; i < EV_COUNT; i++)
    if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
  if (e->ev_cand) (void)hipEventDestroy(e->ev_cand);
  if (e->ev_export_free) (void)hipEventDestroy(e->ev_export_free);
  delete e;
  return SGTD_OK;
}

int sgtd_set_stream(sgtd_handle e, void *hip_stream) {
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e) return SGTD_ERR_INVALID;
  // what is queued on the old stream (copies out of the pinned staging buffer included) finishes there first: the
  // staging bytes are reused as soon as the new stream has been waited for
  CHK(xfer_sync(e));
  e->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return SGTD_OK;
}

int sgtd_set_timing(sgtd_handle e, int enabled) {
  if (e && e->grp) { for (int k = 0; k < sgtd_device_count(e); k++) sgtd_set_timing(sgtd_device_handle(e, k), enabled); return SGTD_OK; }
  if (!e) return SGTD_ERR_INVALID;
  e->timing = enabled != 0;
  return SGTD_OK;
}

int sgtd_current_frame_id(sgtd_handle e, uint32_t *out) {
  if (e && e->grp && out) { *out = multi::current_frame_id(e); return SGTD_OK; }
  if (!e || !out) return SGTD_ERR_INVALID;
  *out = e->current_frame_id;
  return SGTD_OK;
}

int64_t sgtd_max_descs(sgtd_handle e, int n_keypoints) {
  if (e && e->grp) return (int64_t)n_keypoints * ((e->cfg.descriptor_near_num - 1) * (e->cfg.descriptor_near_num - 2) / 2);
  if (!e || n_keypoints < 0) return 0;
  return (int64_t)n_keypoints * e->dc.tpi;
}

int sgtd_build(sgtd_handle e, const float *xyz, const uint32_t *label, int n, sgtd_desc_soa *out,
               int64_t capacity, int64_t *n_out) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::build(e, xyz, label, n, out, capacity, n_out);
  if (!e || !out || !n_out || n < 0 || (n > 0 && (!xyz || !label))) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  *n_out = 0;
  if (n == 0) return SGTD_OK;
  CHK(settle_pending(e));
  {
    // One frame, the reference's call pattern: ONE transfer in (offsets, keypoints, labels in one page-locked block), the
    // build, which writes the count and every descriptor field straight into page-locked host memory (the kernel only ever
    // stores to its output: 0.6 MB over the link inside the kernel instead of a 1.2 MB copy behind it), and one wait — the
    // general path below issues a dozen small copies and three waits.  Frames whose descriptors exceed 4 MB take it.
    const long long cap = (long long)n * e->dc.tpi;
    const size_t in_bytes = 16 + (size_t)n * 16, out_bytes = 16 + (size_t)cap * 136;
    if (out_bytes <= ((size_t)4 << 20) && n >= e->dc.K) {
#ifdef SGTD_EXP_FRAME_LAPS      // host time of the call by part, to stderr (an experiment build)
      struct BLaps {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), t = t0;
        std::string s;
        void lap(const char *what) { const auto n = std::chrono::steady_clock::now(); char b[64]; snprintf(b, sizeof b, " %s %.0f", what, std::chrono::duration<double, std::micro>(n - t).count()); s += b; t = n; }
        ~BLaps() { fprintf(stderr, "[build laps us]%s | all %.0f\n", s.c_str(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
      } blaps;
#define BLAP(x) blaps.lap(x)
#else
#define BLAP(x) do { } while (0)
#endif
      CHK(ensure(e, e->b_in, in_bytes));
      const size_t in_room = (in_bytes + 255) & ~(size_t)255;
      if (e->pin_build_cap < in_room + out_bytes) {
        if (e->pin_build) (void)hipHostFree(e->pin_build);
        e->pin_build = nullptr; e->pin_build_cap = 0;
        void *pp = nullptr;
        HIPCHK(hipHostMalloc(&pp, (in_room + out_bytes) * 2, hipHostMallocDefault));
        e->pin_build = static_cast<char *>(pp);
        e->pin_build_cap = (in_room + out_bytes) * 2;
      }
      char *hin = e->pin_build, *hout = e->pin_build + in_room;
      const long long off2[2] = {0, n};
      std::memcpy(hin, off2, 16);
      std::memcpy(hin + 16, xyz, (size_t)n * 12);
      std::memcpy(hin + 16 + (size_t)n * 12, label, (size_t)n * 4);
      HIPCHK(hipMemcpyAsync(e->b_in.p, hin, in_bytes, hipMemcpyHostToDevice, e->stream));
      BLAP("in");
      char *din = e->b_in.as<char>(), *dout = hout;      // (page-locked memory is device-visible at its own address)
      DescArrays o;
      o.side = reinterpret_cast<double *>(dout + 16); o.angle = o.side + cap * 3; o.center = o.angle + cap * 3;
      o.vertex = reinterpret_cast<float *>(o.center + cap * 3); o.label = reinterpret_cast<int *>(o.vertex + cap * 9);
      o.frame = reinterpret_cast<u32 *>(o.label + cap * 3); o.node_id = reinterpret_cast<int *>(o.frame + cap);
      o.qrec = nullptr;
      CHK(launch_build(e, reinterpret_cast<const float *>(din + 16), reinterpret_cast<const u32 *>(din + 16 + (size_t)n * 12),
                       reinterpret_cast<const long long *>(din), 1, n, e->current_frame_id, 0, o, cap, reinterpret_cast<u32 *>(dout)));
      BLAP("launch");
      HIPCHK(hipStreamSynchronize(e->stream));
      BLAP("out_and_wait");
      const u32 cnt = *reinterpret_cast<const u32 *>(hout);
      *n_out = cnt;
      if ((int64_t)cnt > capacity) return SGTD_ERR_CAPACITY;
      const char *hp = hout + 16;
      auto take = [&](void *dst, size_t width, size_t elem) {       // field of `width` elements per descriptor
        if (dst) std::memcpy(dst, hp, (size_t)cnt * width * elem);
        hp += (size_t)cap * width * elem;
      };
      take(out->side, 3, 8); take(out->angle, 3, 8); take(out->center, 3, 8); take(out->vertex, 9, 4); take(out->label, 3, 4);
      take(out->frame, 1, 4); take(out->node_id, 3, 4);
      BLAP("fields");
#undef BLAP
      return SGTD_OK;
    }
  }
  int64_t off[2] = {0, n};
  const float *dx; const u32 *dl; int max_n;
  CHK(stage_inputs(e, xyz, label, off, 1, 0, &dx, &dl, &max_n, /*own_set=*/true));
  const long long stride = (long long)n * e->dc.tpi;
  CHK(ensure_store(e, e->tmp, (size_t)std::max<long long>(stride, 1)));
  CHK(ensure(e, e->tmp_count, sizeof(u32)));
  CHK(launch_build(e, dx, dl, e->b_kp_off_dev.as<long long>(), 1, max_n, e->current_frame_id, 0,
                   e->tmp.view(), stride, e->tmp_count.as<u32>()));
  u32 cnt = 0;
  CHK(d2h(e, &cnt, e->tmp_count.p, sizeof(u32)));
  CHK(xfer_sync(e));
  *n_out = cnt;
  if ((int64_t)cnt > capacity) return SGTD_ERR_CAPACITY;
  return copy_out(e, e->tmp, 0, cnt, out, 0);
}

int sgtd_add(sgtd_handle e, const sgtd_desc_soa *d, int64_t n) {
  if (e && e->grp) return multi::add(e, d, n);
  if (!e || n < 0 || (n > 0 && (!d || !d->side || !d->label || !d->frame))) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  u32 lo = 0xFFFFFFFFu, hi = 0;
  for (int64_t i = 0; i < n; i++) {
    lo = std::min(lo, d->frame[i]);
    hi = std::max(hi, d->frame[i]);
  }
  if (n > 0 && hi >= (u32)e->cfg.max_frame_n) return SGTD_ERR_FRAME_LIMIT;
  if (e->attached_to) { e->err = "the table belongs to another handle (sgtd_attach_table): add to its owner"; return SGTD_ERR_STATE; }
  CHK(settle_pending(e));
  e->table_version++;
  e->current_frame_id++;  // STDesc.cpp:151, before anything is inserted
  e->n_add_calls++;
  if (n == 0) return SGTD_OK;
  CHK(ensure_store(e, e->tab, (size_t)(e->n_entries + n), true));
  CHK(copy_in(e, e->tab, (size_t)e->n_entries, (size_t)n, d));
  e->n_entries += n;
  note_frames(e, lo, hi);
  e->append_min_frame = std::min(e->append_min_frame, lo);
  e->finalized = false;
  e->batch_valid = false;
  return SGTD_OK;
}

int sgtd_add_frames(sgtd_handle e, const float *xyz, const uint32_t *label, const int64_t *kp_off,
                    int n_frames, int device_ptrs) {
  if (e && e->grp) return multi::add_frames(e, xyz, label, kp_off, n_frames, device_ptrs);
  if (!e || n_frames < 0 || !kp_off) return SGTD_ERR_INVALID;
  if (n_frames == 0) return SGTD_OK;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (e->attached_to) { e->err = "the table belongs to another handle (sgtd_attach_table): add to its owner"; return SGTD_ERR_STATE; }
  if ((uint64_t)e->current_frame_id + (uint64_t)n_frames > (uint64_t)e->cfg.max_frame_n)
    return SGTD_ERR_FRAME_LIMIT;
  CHK(settle_pending(e));
  e->table_version++;
  const int chunk = 512;
  for (int f0 = 0; f0 < n_frames; f0 += chunk) {
    const int nf = std::min(chunk, n_frames - f0);
    const float *dx; const u32 *dl; int max_n;
    CHK(stage_inputs(e, xyz, label, kp_off + f0, nf, device_ptrs, &dx, &dl, &max_n));
    const long long stride = (long long)std::max(max_n, 1) * e->dc.tpi;
    CHK(ensure_store(e, e->tmp, (size_t)stride * nf));
    CHK(ensure(e, e->tmp_count, (size_t)nf * sizeof(u32)));
    CHK(ensure(e, e->cnt_scan, (size_t)nf * sizeof(u32)));
    CHK(launch_build(e, dx, dl, e->kp_off_dev.as<long long>(), nf, max_n, e->current_frame_id, 1,
                     e->tmp.view(), stride, e->tmp_count.as<u32>()));
    std::vector<u32> cnt(nf);
    HIPCHK(hipMemcpyAsync(cnt.data(), e->tmp_count.p, (size_t)nf * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<u32> goff(nf);
    long long total = 0;
    for (int f = 0; f < nf; f++) { goff[f] = (u32)total; total += cnt[f]; }
    if (e->n_entries + total >= (1ll << 32) - 2) return SGTD_ERR_UNSUPPORTED;
    CHK(ensure_store(e, e->tab, (size_t)(e->n_entries + total), true));
    HIPCHK(hipMemcpyAsync(e->cnt_scan.p, goff.data(), (size_t)nf * sizeof(u32), hipMemcpyHostToDevice, e->stream));
    AppendParams A;
    A.in_stride = stride; A.count = e->tmp_count.as<u32>(); A.goff = e->cnt_scan.as<u32>(); A.g0 = e->n_entries;
    append_descs_kernel<<<nf, 256, 0, e->stream>>>(A, e->tmp.view(), e->tab.view());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));  // goff staging vector dies here
    e->n_entries += total;
    note_frames(e, e->current_frame_id, e->current_frame_id + (u32)nf - 1);
    e->append_min_frame = std::min(e->append_min_frame, e->current_frame_id);
    e->current_frame_id += (u32)nf;   // one AddSTDescs (:151) per frame
    e->n_add_calls += nf;
  }
  e->finalized = false;
  e->batch_valid = false;
  return SGTD_OK;
}

int sgtd_attach_table(sgtd_handle v, sgtd_handle o) {
  if (!v || !o || v == o) return SGTD_ERR_INVALID;
  if (v->grp || o->grp) { v->err = "not available on a multi-device handle"; return SGTD_ERR_UNSUPPORTED; }
  sgtd_engine *e = v;
  if (o->attached_to) { e->err = "attach to the handle that owns the table"; return SGTD_ERR_INVALID; }
  if (v->cfg.device_id != o->cfg.device_id || v->cfg.descriptor_near_num != o->cfg.descriptor_near_num ||
      v->cfg.descriptor_min_len != o->cfg.descriptor_min_len || v->cfg.descriptor_max_len != o->cfg.descriptor_max_len ||
      v->cfg.std_side_resolution != o->cfg.std_side_resolution || v->cfg.rough_dis_threshold != o->cfg.rough_dis_threshold ||
      v->cfg.max_frame_n != o->cfg.max_frame_n) {
    e->err = "sgtd_attach_table: the two handles are configured differently";
    return SGTD_ERR_INVALID;
  }
  if (!v->attached_to && (v->n_entries != 0 || v->n_views != 0)) { e->err = "sgtd_attach_table: this handle holds a table of its own"; return SGTD_ERR_STATE; }
  HIPCHK(hipSetDevice(v->cfg.device_id));
  CHK(settle_pending(v));
  HIPCHK(hipStreamSynchronize(v->stream));
  {
    const int st = do_finalize(o);      // the probe layout the view borrows
    if (st != SGTD_OK) { e->err = "sgtd_attach_table: the owner's table could not be finalized"; return st; }
    if (hipStreamSynchronize(o->stream) != hipSuccess) return SGTD_ERR_HIP;
  }
  auto borrow = [](DevBuf &dst, const DevBuf &src) {
    if (dst.p && !dst.borrowed) (void)hipFree(dst.p);
    dst = src;
    dst.borrowed = true;
  };
  borrow(v->tab.side, o->tab.side); borrow(v->tab.angle, o->tab.angle); borrow(v->tab.center, o->tab.center);
  borrow(v->tab.vertex, o->tab.vertex); borrow(v->tab.label, o->tab.label); borrow(v->tab.frame, o->tab.frame);
  borrow(v->tab.node_id, o->tab.node_id);
  v->tab.cap = o->tab.cap;
  for (int k = 0; k < 2; k++) {
    sgtd_engine::Segment &d = v->seg[k];
    const sgtd_engine::Segment &s = o->seg[k];
    borrow(d.hot, s.hot); borrow(d.perm, s.perm); borrow(d.hash, s.hash); borrow(d.bucket_start, s.bucket_start);
    borrow(d.bucket_key, s.bucket_key); borrow(d.dir, s.dir);
    d.hash_mask = s.hash_mask; d.n_buckets = s.n_buckets; d.g0 = s.g0; d.g1 = s.g1; d.sum_len_sq = s.sum_len_sq;
    d.frame_hi = s.frame_hi; d.built = s.built;
  }
  borrow(v->frame_first, o->frame_first); borrow(v->by_frame, o->by_frame); borrow(v->id_of_g, o->id_of_g);
  v->n_seg = o->n_seg; v->id_bits = o->id_bits; v->id_by_frame = o->id_by_frame;
  v->n_entries = o->n_entries; v->n_add_calls = o->n_add_calls; v->have_frames = o->have_frames;
  v->frame_lo = o->frame_lo; v->frame_hi = o->frame_hi; v->current_frame_id = o->current_frame_id;
  v->ms_finalize = o->ms_finalize;
  v->finalized = true;
  v->batch_valid = false;
  if (v->attached_to != o) {
    if (v->attached_to) v->attached_to->n_views--;
    o->n_views++;
  }
  v->attached_to = o;
  v->attached_version = o->table_version;
  return SGTD_OK;
}

int sgtd_finalize(sgtd_handle e) {
  if (e && e->grp) return multi::finalize(e);
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  return do_finalize(e);
}

int sgtd_query_frames(sgtd_handle e, const float *xyz, const uint32_t *label, const int64_t *kp_off,
                      int n_queries, int device_ptrs) {
  if (e && e->grp) return multi::query_frames(e, xyz, label, kp_off, n_queries, device_ptrs);
  if (!e || n_queries <= 0 || !kp_off || !xyz || !label) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(settle_tail(e));
  const float *dx; const u32 *dl; int max_n;
  CHK(stage_inputs(e, xyz, label, kp_off, n_queries, device_ptrs, &dx, &dl, &max_n));
  e->nq = n_queries;
  e->q_stride = (long long)std::max(max_n, 1) * e->dc.tpi;
  e->last_kind = 1; e->last_xyz = dx; e->last_label = dl; e->last_max_n = max_n;
  e->last_qframe = e->current_frame_id;
  e->diag = false;   // a new batch runs the product sweep; sgtd_result_rough re-runs it in the diagnostic form
  e->rec_rate_cap = e->stats.overflowed ? e->rec_rate_cap : std::min<u32>(256, e->rec_rate_cap * 2);   // (a cap a re-run needed recovers slowly)
  CHK(ensure_store(e, e->qd, (size_t)e->q_stride * n_queries));
  CHK(ensure(e, e->q_count, (size_t)n_queries * sizeof(u32)));
  if (!e->rec_cap_fixed && e->stats.last_queries == 0) {
    // first batch of this handle: size the match-record buffer from the table statistics instead
    // of growing it through overflow re-runs (each costs a whole sweep)
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // ... plus what the slabs of all resident streams leave unused: each holds about half a slab of eight expected lists
    const double per_query = est_matches_per_query(e, max_n);
    const double per_desc = per_query / std::max(1.0, 0.62 * (double)max_n * e->dc.tpi);
    const double slab_slack = (double)e->n_cus * 32.0 * SGTD_PAIR * 0.5 * 8.0 * (3.0 * per_desc + 256.0);
    const double margin = e->n_entries > 100000000 ? 1.75 : 1.3;    // (long buckets: the estimate runs low at 100 000 frames)
    const double want = margin * per_query * (double)n_queries + std::min(slab_slack, margin * per_query * (double)n_queries);
    const double cap_mem = (double)free_b / 4.0 / 16.0;    // records + compact list + pairs, a quarter of what is free
    const size_t cap = (size_t)std::min(std::min(want, cap_mem), (double)kRecLimit);
    if (cap > e->rec_cap) e->rec_cap = cap;
    if (std::min(cap / 2, kIndexLimit) > e->pair_cap) e->pair_cap = std::min(cap / 2, kIndexLimit);   // candidate pairs: 0.2 .. 0.5 of the matches
  }
  return enqueue_frames(e);
}

int sgtd_query_descs(sgtd_handle e, const sgtd_desc_soa *q, int64_t nq) {
  if (e && e->grp) return multi::query_descs(e, q, nq);
  if (!e || nq < 0 || (nq > 0 && (!q || !q->side || !q->label || !q->frame))) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(settle_tail(e));
  e->nq = 1;
  e->q_stride = std::max<long long>(nq, 1);
  e->last_kind = 2;
  e->diag = false;
  e->rec_rate_cap = e->stats.overflowed ? e->rec_rate_cap : std::min<u32>(256, e->rec_rate_cap * 2);
  CHK(ensure_store(e, e->qd, (size_t)e->q_stride));
  CHK(ensure(e, e->q_count, sizeof(u32)));
  CHK(copy_in(e, e->qd, 0, (size_t)nq, q));
  e->thr2_pending = nq;      // the descriptors' sweep records (thresholds, gate masks): launch_select writes them — one frame per call in its one launch
  u32 cnt = (u32)nq;
  CHK(h2d(e, e->q_count.p, &cnt, sizeof(u32)));
  CHK(xfer_sync(e));
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_START], e->stream));
  return launch_select(e);
}

int sgtd_max_batch(sgtd_handle e, int n_keypoints, int64_t *max_queries) {
  if (e && e->grp) return multi::max_batch(e, n_keypoints, max_queries);
  if (!e || !max_queries || n_keypoints < 0) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(settle_tail(e));
  const double per_query = std::max(1.0, est_matches_per_query(e, n_keypoints)) * 2.0;   // margin: slab slack, variation
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  // records are named by granules (6.9e10 of them), a batch's candidate pairs one by one: 0.2 .. 0.5 of the matches on the
  // uniform maps (four tenths before anything was measured; per_query carries a factor of two already)
  double pair_share = 0.4;
  if (e->stats.last_M > 0 && e->stats.last_cand_pairs > 0)      // measured on the batch before, half as much again
    pair_share = std::min(1.0, std::max(0.05, 1.5 * (double)e->stats.last_cand_pairs / (double)e->stats.last_M));
  // memory: 4 B per record and 8 B per candidate pair on top of what the buffers already hold (a batch this large takes the
  // per-query passes: no compact lists between the block passes, which would be 8 B per record more)
  const double mem_records = ((double)free_b * 0.8 + (double)e->rec.bytes + (double)e->c_pair.bytes + (double)e->pairs.bytes) / (4.0 + 8.0 * pair_share);
  const double lim = std::min(std::min((double)kRecLimit, mem_records), (double)kIndexLimit / pair_share * 2.0);
  *max_queries = (int64_t)std::max(1.0, std::floor(lim / per_query));
  return SGTD_OK;
}

int sgtd_sync(sgtd_handle e) {
  if (e && e->grp) return multi::sync(e);
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->batch_valid) {
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGTD_OK;
  }
  return sync_batch(e);
}

int sgtd_result_candidates(sgtd_handle e, int32_t *n_cand, int32_t *cand_frame, int32_t *cand_votes,
                           int64_t *pair_off) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::result_candidates(e, n_cand, cand_frame, cand_votes, pair_off);
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(sync_batch(e));
  const int nq = e->nq, cn = e->dc.cand_num;
  if (n_cand) std::memcpy(n_cand, e->h_n_cand.data(), nq * sizeof(int));
  if (cand_frame) std::memcpy(cand_frame, e->h_cand_frame.data(), (size_t)nq * cn * sizeof(int));
  if (cand_votes) std::memcpy(cand_votes, e->h_cand_votes.data(), (size_t)nq * cn * sizeof(int));
  if (pair_off)
    for (size_t i = 0; i < (size_t)nq * (cn + 1); i++) pair_off[i] = e->h_pair_off[i];
  return SGTD_OK;
}

int sgtd_export_candidates_dev(sgtd_handle e, int32_t *d_cand_frame, int32_t *d_cand_votes) {
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e || !d_cand_frame || !d_cand_votes) return SGTD_ERR_INVALID;
  if (!e->batch_valid) return SGTD_ERR_STATE;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  // a batch that outgrew a work buffer has empty candidate tables until it is re-run:
  // resolve that first (one stream wait per batch; the tables of an intact batch are final)
  CHK(sync_batch(e));
  const size_t bytes = (size_t)e->nq * e->dc.cand_num * sizeof(int);
  HIPCHK(hipMemcpyAsync(d_cand_frame, e->cand_frame.p, bytes, hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(d_cand_votes, e->cand_votes.p, bytes, hipMemcpyDeviceToDevice, e->stream));
  return SGTD_OK;
}

int64_t sgtd_candidate_export_ints(int n_queries, int cand_num) {
  if (n_queries < 0 || cand_num < 0) return 0;
  return (int64_t)xchg_packed_ints(n_queries, cand_num);
}

#define SGTD_NO_GROUP(e)                                                                                                   \
  if ((e) && (e)->grp) { (e)->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }

int sgtd_set_candidate_export(sgtd_handle e, int32_t *d_packed, int64_t capacity_ints) {
  SGTD_NO_GROUP(e);
  if (!e || capacity_ints < 0 || (d_packed && capacity_ints < SGTD_XCHG_FLAG_WORDS)) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  // (what is enqueued may still write the buffer registered before)
  HIPCHK(hipStreamSynchronize(e->stream));
  e->export_packed = d_packed;
  e->export_cap = d_packed ? (size_t)capacity_ints : 0;
  e->export_busy = false;
  return SGTD_OK;
}

int sgtd_export_wait(sgtd_handle e, void *side_stream) {
  SGTD_NO_GROUP(e);
  if (!e) return SGTD_ERR_INVALID;
  if (!e->export_packed || !e->batch_valid) return SGTD_ERR_STATE;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  HIPCHK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(side_stream), e->ev_cand, 0));
  return SGTD_OK;
}

int sgtd_export_release(sgtd_handle e, void *side_stream) {
  SGTD_NO_GROUP(e);
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  HIPCHK(hipEventRecord(e->ev_export_free, reinterpret_cast<hipStream_t>(side_stream)));
  e->export_busy = true;
  return SGTD_OK;
}

int sgtd_merge_candidates_dev(sgtd_handle e, void *stream, const int32_t *d_gathered, int n_tables, int my_table, int n_queries,
                              int32_t *d_frame, int32_t *d_votes, int32_t *d_n_cand, int32_t *d_src, uint64_t *d_keep, int32_t *d_flags) {
  SGTD_NO_GROUP(e);
  if (!e || !d_gathered || !d_frame || !d_votes || !d_n_cand || !d_src || !d_keep || !d_flags || n_tables < 1 || n_queries < 0 ||
      my_table < -1 || my_table >= n_tables)
    return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num;
  if ((long long)n_tables * cn > (long long)SGTD_MERGE_PER_LANE * SGTD_WAVE || n_tables > 0x7FFFFF) return SGTD_ERR_UNSUPPORTED;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  // (min_votes 5: max_vote >= 5, STDesc.cpp:433)
  const int n_keys = n_tables * cn;
#define SGTD_MERGE(PER) merge_candidates_kernel<PER><<<std::max(1, grid_for(n_queries, 4)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>( \
      d_gathered, (long long)xchg_packed_ints(n_queries, cn), n_tables, my_table, n_queries, cn, 5, d_frame, d_votes, d_n_cand, d_src,             \
      reinterpret_cast<u64 *>(d_keep), d_flags)
  if (n_keys <= SGTD_WAVE) SGTD_MERGE(1);
  else if (n_keys <= 4 * SGTD_WAVE) SGTD_MERGE(4);
  else if (n_keys <= 8 * SGTD_WAVE) SGTD_MERGE(8);
  else SGTD_MERGE(SGTD_MERGE_PER_LANE);
#undef SGTD_MERGE
  HIPCHK(hipGetLastError());
  return SGTD_OK;
}

int sgtd_gather_verified_dev(sgtd_handle e, void *stream, const double *d_gathered, int n_tables, const int32_t *d_src, int n_queries,
                             double *d_score, double *d_pose) {
  SGTD_NO_GROUP(e);
  if (!e || !d_gathered || !d_src || !d_score || !d_pose || n_tables < 1 || n_queries < 0) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num;
  if (n_queries == 0) return SGTD_OK;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  gather_verified_kernel<<<grid_for((long long)n_queries * cn, 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
      d_gathered, (long long)n_queries * cn * 13, d_src, n_queries, cn, d_score, d_pose);
  HIPCHK(hipGetLastError());
  return SGTD_OK;
}

int sgtd_set_deferred_lists(sgtd_handle e, int on) {
  SGTD_NO_GROUP(e);
  if (!e) return SGTD_ERR_INVALID;
  e->defer_lists = on != 0;
  return SGTD_OK;
}

int sgtd_finish_lists(sgtd_handle e, const uint64_t *d_keep) {
  SGTD_NO_GROUP(e);
  if (!e) return SGTD_ERR_INVALID;
  if (!e->batch_valid) return SGTD_ERR_STATE;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->pairs_per_query) {
    // the lists of this batch came from the passes of one wave per 128-descriptor block (small batches, frame spans
    // beyond LDS): they hold every local candidate already; a mask only matters to sgtd_verify_masked
    return SGTD_OK;
  }
  return launch_lists(e, make_views(e), reinterpret_cast<const u64 *>(d_keep), /*first=*/false);
}
#undef SGTD_NO_GROUP

int sgtd_result_query_desc_count(sgtd_handle e, int q, int64_t *n) {
  if (e && e->grp) return multi::result_query_desc_count(e, q, n);
  if (!e || !n) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(sync_batch(e));
  if (q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  *n = e->h_count[q];
  return SGTD_OK;
}

int sgtd_result_pairs(sgtd_handle e, int q, int32_t *q_idx, int64_t *db_entry, int64_t capacity,
                      int64_t *n_pairs) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::result_pairs(e, q, q_idx, db_entry, capacity, n_pairs);
  if (!e || !n_pairs) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(sync_batch(e));
  if (q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num;
  const int64_t n = e->h_pair_off[(size_t)q * (cn + 1) + cn];
  *n_pairs = n;
  if (n > capacity) return SGTD_ERR_CAPACITY;
  if (n == 0) return SGTD_OK;
  std::vector<u64> pr(n);
  CHK(d2h(e, pr.data(), e->pairs.as<u64>() + e->h_pair_base[q], n * sizeof(u64)));
  CHK(xfer_sync(e));
  for (int64_t i = 0; i < n; i++) {
    if (q_idx) q_idx[i] = (int32_t)(pr[i] >> 32);
    if (db_entry) db_entry[i] = (int64_t)(pr[i] & 0xFFFFFFFFull);
  }
  return SGTD_OK;
}

int sgtd_result_query_descs(sgtd_handle e, int q, sgtd_desc_soa *out, int64_t capacity, int64_t *n_out) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::result_query_descs(e, q, out, capacity, n_out);
  if (!e || !out || !n_out) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(sync_batch(e));
  if (q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  *n_out = e->h_count[q];
  if (*n_out > capacity) return SGTD_ERR_CAPACITY;
  return copy_out(e, e->qd, (size_t)q * e->q_stride, e->h_count[q], out, 0);
}

int sgtd_result_votes(sgtd_handle e, int q, uint32_t *votes, int64_t capacity, uint32_t *frame_lo, int64_t *n) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::result_votes(e, q, votes, capacity, frame_lo, n);
  if (!e || !n) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(sync_batch(e));
  if (q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  const u32 span = e->have_frames ? (e->frame_hi - e->frame_lo + 1) : 1;
  *n = span;
  if (frame_lo) *frame_lo = e->have_frames ? e->frame_lo : 0;
  if (!votes) return SGTD_OK;
  if ((int64_t)span > capacity) return SGTD_ERR_CAPACITY;
  HIPCHK(hipMemcpyAsync(votes, e->votes.as<u32>() + (size_t)q * span, (size_t)span * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return SGTD_OK;
}

int sgtd_result_rough(sgtd_handle e, int q, int32_t *q_idx, int32_t *cell, int64_t *db_entry, uint32_t *frame,
                      double *dis, int64_t capacity, int64_t *n_rough) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e || !n_rough) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->batch_valid) return SGTD_ERR_STATE;
  if (!e->diag) {
    // the ordered rough list (cell index, distance, reference order inside a cell) comes from the
    // diagnostic probe build: switch it on and re-run the batch
    e->diag = true;
    CHK(rec_alloc(e, false));     // (the diagnostic arrays; the re-run below sizes the rest)
    CHK(rerun(e));
  }
  CHK(sync_batch(e));
  if (q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  const int64_t n = e->h_q_M[q];
  *n_rough = n;
  if (n > capacity) return SGTD_ERR_CAPACITY;
  if (n == 0) return SGTD_OK;
  CHK(ensure(e, e->rough_qi, n * sizeof(u32)));
  CHK(ensure(e, e->rough_entry, n * sizeof(u32)));
  CHK(ensure(e, e->rough_frame, n * sizeof(u32)));
  if (e->diag) {
    CHK(ensure(e, e->rough_cell, n));
    CHK(ensure(e, e->rough_dis, n * sizeof(double)));
  }
  Views v = make_views(e);
  rough_gather_kernel<<<1, 256, 0, e->stream>>>(v.Q, v.B, v.T.map, q, e->rough_qi.as<u32>(), e->rough_entry.as<u32>(),
                                                 e->rough_frame.as<u32>(),
                                                 e->diag ? e->rough_cell.as<unsigned char>() : nullptr,
                                                 e->diag ? e->rough_dis.as<double>() : nullptr);
  HIPCHK(hipGetLastError());
  std::vector<u32> a(n), g(n);
  std::vector<unsigned char> c(n);
  HIPCHK(hipMemcpyAsync(a.data(), e->rough_qi.p, n * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(g.data(), e->rough_entry.p, n * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
  if (cell) HIPCHK(hipMemcpyAsync(c.data(), e->rough_cell.p, n, hipMemcpyDeviceToHost, e->stream));
  if (frame) HIPCHK(hipMemcpyAsync(frame, e->rough_frame.p, n * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
  if (dis) HIPCHK(hipMemcpyAsync(dis, e->rough_dis.p, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int64_t i = 0; i < n; i++) {
    if (q_idx) q_idx[i] = (int32_t)a[i];
    if (db_entry) db_entry[i] = (int64_t)g[i];
    if (cell) cell[i] = c[i];
  }
  return SGTD_OK;
}

int sgtd_verify_masked(sgtd_handle e, const uint64_t *d_keep) {
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e) return SGTD_ERR_INVALID;
  e->verify_keep = reinterpret_cast<const u64 *>(d_keep);
  const int st = sgtd_verify(e);
  e->verify_keep = nullptr;
  return st;
}

}  // extern "C"

namespace {
// candidate_verify + triangle_solver (STDesc.cpp:462-571) of every (query, candidate) of the batch, enqueued behind it;
// total = an upper bound on the pairs of all lists (sizes the per-pair scratch); guard: the kernels look at the batch's
// overflow flags first (the batch has not been waited for)
int verify_enqueue(sgtd_engine *e, int64_t total, bool guard) {
  const int cn = e->dc.cand_num, nq = e->nq;
  CHK(ensure(e, e->v_score, (size_t)nq * cn * sizeof(double)));
  CHK(ensure(e, e->v_pose, (size_t)nq * cn * 12 * sizeof(double)));
  CHK(ensure(e, e->v_inlier, (size_t)std::max<int64_t>(total, 1)));
  VerifyParams P;
  P.pairs = e->pairs.as<u64>(); P.pair_off = e->pair_off.as<long long>(); P.q_pair_base = e->q_pair_base.as<u32>();
  P.n_cand = e->n_cand.as<int>(); P.cand_num = cn; P.q_stride = e->q_stride;
  P.q_vertex = e->qd.vertex.as<float>(); P.q_center = e->qd.center.as<double>();
  P.t_vertex = e->tab.vertex.as<float>(); P.t_center = e->tab.center.as<double>();
  P.score = e->v_score.as<double>(); P.pose = e->v_pose.as<double>(); P.inlier = e->v_inlier.as<unsigned char>();
  P.thr2 = 9.0;   // sqrt_rn(y) < 3.0 <=> y < 9.0 (sqrt(9) = 3, sqrt(pred(9)) rounds to pred(3)); dis_threshold :469
  { const char *o = getenv("SGTD_VERIFY_EXACT"); P.exact_only = (o && atoi(o)) ? 1 : 0; }
  // the vote pass: on the matrix cores (verify_mfma.hip.h) unless SGTD_VERIFY_FORM=valu asks for the packed-f32 form
  bool mfma = true;
  { const char *o = getenv("SGTD_VERIFY_FORM"); if (o && !strcmp(o, "valu")) mfma = false; }
  CHK(ensure(e, e->v_hyp64, (size_t)nq * cn * SGTD_VERIFY_MAX_HYP * SGTD_HYP_F64 * sizeof(double)));
  CHK(ensure(e, e->v_bound, (size_t)nq * cn * 2 * sizeof(u32)));
  P.hyp64 = e->v_hyp64.as<double>(); P.bound = e->v_bound.as<u32>();
  P.passed = nullptr; P.hyp32 = nullptr; P.words = nullptr; P.hypB = nullptr; P.tau = nullptr;
  if (mfma) {
    // a row of 64 vote words per 32 pairs; candidate k of the batch starts at row (first pair / 32 + k)
    CHK(ensure(e, e->v_words, ((size_t)std::max<int64_t>(total, 1) / 32 + (size_t)nq * cn + 2) * 64 * sizeof(u32)));
    CHK(ensure(e, e->v_hypB, (size_t)nq * cn * 6 * 64 * sizeof(uint4)));
    CHK(ensure(e, e->v_tau, (size_t)nq * cn * 2 * SGTD_VERIFY_MAX_HYP * sizeof(float)));
    P.words = e->v_words.as<u32>(); P.hypB = e->v_hypB.as<uint4>(); P.tau = e->v_tau.as<float>();
  } else {
    // per-pair vote masks between the kernel's two passes (a buffer of their own: the compact candidate lists in c_pair may
    // still be needed by a re-run of the batch's list pass when this is enqueued behind an unfinished batch)
    CHK(ensure(e, e->v_words, (size_t)std::max<int64_t>(total, 1) * sizeof(u64)));
    CHK(ensure(e, e->v_hyp32, (size_t)nq * cn * SGTD_VERIFY_MAX_HYP * SGTD_HYP_F32 * sizeof(float)));
    P.passed = e->v_words.as<u64>(); P.hyp32 = e->v_hyp32.as<float>();
    HIPCHK(hipMemsetAsync(e->v_pose.p, 0, (size_t)nq * cn * 12 * sizeof(double), e->stream));     // (the matrix-core kernel writes every pose itself)
  }
  P.keep = e->verify_keep;
  P.inl_count = nullptr;
  if (guard && mfma) {        // (sgtd_search_frame: the kernel leaves the candidates' inlier counts behind)
    CHK(ensure(e, e->inl_counts, (size_t)std::max(nq * cn, SGTD_MAX_CAND) * sizeof(u32)));
    P.inl_count = e->inl_counts.as<u32>();
  }
  e->verify_counted = P.inl_count != nullptr;
  P.overflow = guard ? reinterpret_cast<const int *>(e->cursors.as<u32>() + 10) : nullptr;
  P.order = nullptr; P.n_blocks = (u32)(nq * cn);
  int grid = nq * cn;
  // a batch of many candidates is dispatched in the order of the candidates' frames (SGTD_VERIFY_ORDER=0: as they stand)
  static const bool order_on = [] { const char *o = getenv("SGTD_VERIFY_ORDER"); return !(o && !atoi(o)); }();
  if (mfma && order_on && !guard && nq * cn >= 4096 && e->have_frames) {
    const u32 nb = (u32)(nq * cn);
    for (int k = 0; k < 2; k++) { CHK(ensure(e, e->v_okey[k], (size_t)nb * sizeof(u32))); CHK(ensure(e, e->v_oval[k], (size_t)nb * sizeof(u32))); }
    u32 *kin = e->v_okey[0].as<u32>(), *kout = e->v_okey[1].as<u32>(), *vin = e->v_oval[0].as<u32>(), *vout = e->v_oval[1].as<u32>();
    const u32 last = e->frame_hi + 1u;          // key of a candidate slot that is empty
    verify_order_keys_kernel<<<grid_for(nb, 256), 256, 0, e->stream>>>(e->cand_frame.as<int>(), e->n_cand.as<int>(), cn, nb, last, kin, vin);
    HIPCHK(hipGetLastError());
    int bits = 1;
    while (bits < 32 && (last >> bits)) bits++;
    CHK(radix_sort_pairs<u32>(e, kin, kout, vin, vout, (long long)nb, bits, false));
    P.order = vin;
    grid = 8 * (int)((nb + 7) / 8);
  }
  verify_solve_kernel<<<nq * cn, SGTD_WAVE, 0, e->stream>>>(P);
  HIPCHK(hipGetLastError());
  if (mfma && nq * cn >= 1024) verify_mfma_kernel<4><<<grid, 4 * SGTD_WAVE, 0, e->stream>>>(P);
  else if (mfma) verify_mfma_kernel<8><<<grid, 8 * SGTD_WAVE, 0, e->stream>>>(P);
  else verify_kernel<<<nq * cn, SGTD_VERIFY_THREADS, 0, e->stream>>>(P);
  HIPCHK(hipGetLastError());
#ifdef SGTD_EXP_VSTAT
  {
    unsigned long long st[8];
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_vstat), sizeof(st)));
    fprintf(stderr, "[vstat] steps %llu  steps with a pair left by vertex A %llu  with eight or more %llu | (pair, hypothesis) tests %llu  left by A %llu  of them in steps "
            "with eight or more %llu  not far on all three %llu  certain votes %llu\n", st[0], st[1], st[3], st[7], st[2], st[5], st[6], st[4]);
    unsigned long long z[8] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_vstat), z, sizeof(z)));
    HIPCHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_vmstat), sizeof(st)));
    fprintf(stderr, "[vmstat] (32 pairs x 32 hypotheses) steps %llu  with something left open %llu | combinations queued for the exact test %llu  of them votes %llu  "
            "queue drains %llu  certain votes %llu\n", st[0], st[1], st[2], st[3], st[4], st[5]);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_vmstat), z, sizeof(z)));
  }
#endif
  return SGTD_OK;
}
}  // namespace

extern "C" {

int sgtd_verify(sgtd_handle e) {
  if (e && e->grp) return multi::verify(e);
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(sync_batch(e));
  if (!e->batch_valid) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num, nq = e->nq;
  if (nq == 0) { e->verified = true; return SGTD_OK; }
  int64_t total = 0;
  for (int q = 0; q < nq; q++) total = std::max<int64_t>(total, (int64_t)e->h_pair_base[q] + e->h_pair_off[(size_t)q * (cn + 1) + cn]);
  CHK(verify_enqueue(e, total, /*guard=*/false));
  e->verified = true;
  return SGTD_OK;
}

int sgtd_result_verify(sgtd_handle e, int q, double *score, double *pose) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::result_verify(e, q, score, pose);
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->verified || !e->batch_valid || q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num;
  if (score) CHK(d2h(e, score, e->v_score.as<double>() + (size_t)q * cn, cn * sizeof(double)));
  if (pose) CHK(d2h(e, pose, e->v_pose.as<double>() + (size_t)q * cn * 12, (size_t)cn * 12 * sizeof(double)));
  CHK(xfer_sync(e));
  return SGTD_OK;
}

int sgtd_export_verify_dev(sgtd_handle e, double *d_score, double *d_pose) {
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->verified || !e->batch_valid) return SGTD_ERR_STATE;
  const size_t n = (size_t)e->nq * e->dc.cand_num;
  if (n == 0) return SGTD_OK;
  if (d_score) HIPCHK(hipMemcpyAsync(d_score, e->v_score.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  if (d_pose) HIPCHK(hipMemcpyAsync(d_pose, e->v_pose.p, n * 12 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  return SGTD_OK;
}

int sgtd_result_inliers(sgtd_handle e, int q, int cand, int32_t *idx, int64_t capacity, int64_t *n) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::result_inliers(e, q, cand, idx, capacity, n);
  if (!e || !n) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->verified || !e->batch_valid || q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num;
  if (cand < 0 || cand >= e->h_n_cand[q]) return SGTD_ERR_INVALID;
  const int64_t lo = e->h_pair_off[(size_t)q * (cn + 1) + cand], hi = e->h_pair_off[(size_t)q * (cn + 1) + cand + 1];
  std::vector<unsigned char> fl((size_t)(hi - lo));
  if (hi > lo) {
    CHK(d2h(e, fl.data(), e->v_inlier.as<unsigned char>() + e->h_pair_base[q] + lo, (size_t)(hi - lo)));
    CHK(xfer_sync(e));
  }
  int64_t cnt = 0;
  for (int64_t j = 0; j < hi - lo; j++)
    if (fl[(size_t)j]) {
      if (idx && cnt < capacity) idx[cnt] = (int32_t)j;
      cnt++;
    }
  *n = cnt;
  return (idx && cnt > capacity) ? SGTD_ERR_CAPACITY : SGTD_OK;
}

int sgtd_result_inlier_pairs(sgtd_handle e, int q, int64_t *cand_off, int32_t *q_idx, int64_t *db_entry, int64_t capacity,
                             int64_t *n_pairs) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::result_inlier_pairs(e, q, cand_off, q_idx, db_entry, capacity, n_pairs);
  if (!e || !n_pairs || !cand_off) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->verified || !e->batch_valid || q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num;
  const int64_t total = e->h_pair_off[(size_t)q * (cn + 1) + cn];
  CHK(ensure(e, e->inl_pairs, (size_t)std::max<int64_t>(total, 1) * sizeof(u64)));
  CHK(ensure(e, e->inl_off, (size_t)(cn + 1) * sizeof(long long)));
  inlier_pairs_kernel<<<1, SGTD_INLIER_THREADS, 0, e->stream>>>(e->pairs.as<u64>() + e->h_pair_base[q], e->v_inlier.as<unsigned char>() + e->h_pair_base[q],
                                                                e->pair_off.as<long long>() + (size_t)q * (cn + 1), cn, e->inl_pairs.as<u64>(),
                                                                e->inl_off.as<long long>());
  HIPCHK(hipGetLastError());
  std::vector<long long> off((size_t)cn + 1);
  CHK(d2h(e, off.data(), e->inl_off.p, off.size() * sizeof(long long)));
  CHK(xfer_sync(e));
  for (int k = 0; k <= cn; k++) cand_off[k] = off[(size_t)k];
  const int64_t n = off[(size_t)cn];
  *n_pairs = n;
  if (n > capacity) return SGTD_ERR_CAPACITY;
  if (n == 0) return SGTD_OK;
  std::vector<u64> pr((size_t)n);
  CHK(d2h(e, pr.data(), e->inl_pairs.p, (size_t)n * sizeof(u64)));
  CHK(xfer_sync(e));
  for (int64_t i = 0; i < n; i++) {
    if (q_idx) q_idx[i] = (int32_t)(pr[(size_t)i] >> 32);
    if (db_entry) db_entry[i] = (int64_t)(pr[(size_t)i] & 0xFFFFFFFFull);
  }
  return SGTD_OK;
}

int sgtd_result_inlier_entries(sgtd_handle e, int q, int64_t *cand_off, int32_t *q_idx, sgtd_desc_soa *entries, int64_t capacity,
                               int64_t *n_pairs) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (!e || !n_pairs || !cand_off) return SGTD_ERR_INVALID;
  if (e->grp) {
    // several devices: the pairs from the devices that own the candidates, then their entries (db_entry ids name the owner)
    std::vector<int64_t> ent((size_t)std::max<int64_t>(capacity, 0));
    const int st = sgtd_result_inlier_pairs(e, q, cand_off, q_idx, ent.data(), capacity, n_pairs);
    if (st != SGTD_OK) return st;
    return entries ? sgtd_fetch_entries(e, ent.data(), *n_pairs, entries) : SGTD_OK;
  }
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(view_current(e));
  if (!e->verified || !e->batch_valid || q < 0 || q >= e->nq) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num;
  const int64_t total = e->h_pair_off[(size_t)q * (cn + 1) + cn];
  CHK(ensure(e, e->inl_pairs, (size_t)std::max<int64_t>(total, 1) * sizeof(u64)));
  CHK(ensure(e, e->inl_off, (size_t)(cn + 1) * sizeof(long long)));
  // index list, q_idx list and the gathered entries for the most there can be, so that everything is enqueued
  // before the one wait for the offsets
  CHK(ensure(e, e->fetch_idx, (size_t)std::max<int64_t>(total, 1) * (sizeof(long long) + sizeof(int))));
  CHK(ensure_store(e, e->fetch, (size_t)std::max<int64_t>(total, 1)));
  long long *d_idx = e->fetch_idx.as<long long>();
  int *d_qi = reinterpret_cast<int *>(d_idx + std::max<int64_t>(total, 1));
  inlier_pairs_kernel<<<1, SGTD_INLIER_THREADS, 0, e->stream>>>(e->pairs.as<u64>() + e->h_pair_base[q], e->v_inlier.as<unsigned char>() + e->h_pair_base[q],
                                                                e->pair_off.as<long long>() + (size_t)q * (cn + 1), cn, e->inl_pairs.as<u64>(),
                                                                e->inl_off.as<long long>());
  HIPCHK(hipGetLastError());
  if (total > 0) {
    split_pairs_kernel<<<grid_for(total, 256), 256, 0, e->stream>>>(e->inl_pairs.as<u64>(), e->inl_off.as<long long>() + cn, (long long)total, d_idx, d_qi);
    HIPCHK(hipGetLastError());
    gather_entries_counted_kernel<<<grid_for(total, 256), 256, 0, e->stream>>>(d_idx, e->inl_off.as<long long>() + cn, (long long)total, e->tab.view(), e->fetch.view());
    HIPCHK(hipGetLastError());
  }
  std::vector<long long> off((size_t)cn + 1);
  CHK(d2h(e, off.data(), e->inl_off.p, off.size() * sizeof(long long)));
  CHK(xfer_sync(e));
  for (int k = 0; k <= cn; k++) cand_off[k] = off[(size_t)k];
  const int64_t n = off[(size_t)cn];
  *n_pairs = n;
  if (n > capacity) return SGTD_ERR_CAPACITY;
  if (n == 0) return SGTD_OK;
  if (q_idx) CHK(d2h(e, q_idx, d_qi, (size_t)n * sizeof(int)));
  if (entries) return copy_out(e, e->fetch, 0, (size_t)n, entries, 0);
  return xfer_sync(e);
}

int sgtd_search_loop(sgtd_handle e, double icp_threshold, int32_t *best_cand, int32_t *best_frame, double *best_score) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::search_loop(e, icp_threshold, best_cand, best_frame, best_score);
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (!e->verified || !e->batch_valid) return SGTD_ERR_INVALID;
  const int cn = e->dc.cand_num, nq = e->nq;
  if (nq == 0) return SGTD_OK;
  CHK(ensure(e, e->v_best, (size_t)nq * (2 * sizeof(int) + sizeof(double))));
  double *d_score = e->v_best.as<double>();
  int *d_cand = reinterpret_cast<int *>(d_score + nq), *d_frame = d_cand + nq;
  search_loop_kernel<<<grid_for(nq, 256), 256, 0, e->stream>>>(e->v_score.as<double>(), e->cand_frame.as<int>(), e->n_cand.as<int>(), cn, nq,
                                                               icp_threshold, d_cand, d_frame, d_score);
  HIPCHK(hipGetLastError());
  if (best_cand) CHK(d2h(e, best_cand, d_cand, nq * sizeof(int)));
  if (best_frame) CHK(d2h(e, best_frame, d_frame, nq * sizeof(int)));
  if (best_score) CHK(d2h(e, best_score, d_score, nq * sizeof(double)));
  CHK(xfer_sync(e));
  return SGTD_OK;
}

// [p, p + bytes) is page-locked host memory a kernel can write through the same address (hipHostMalloc / sgtd_host_alloc)
static bool device_can_write(const void *p, size_t bytes) {
  if (!p) return true;        // (a member the caller does not want)
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (a.type != hipMemoryTypeHost || a.devicePointer != p) return false;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, const_cast<void *>(p)) != hipSuccess) { (void)hipGetLastError(); return false; }
  return static_cast<const char *>(p) + bytes <= static_cast<const char *>(base) + size;
}

// One query frame through candidate_selector, candidate_verify and the inlier pairs with their table entries in ONE call
// and (normally) two waits: the reference's per-frame call pattern (semantic_graph_localization.cpp:590-603 ->
// STDesc.cpp:84-147) as sgtd_query_descs + sgtd_verify + sgtd_result_candidates + sgtd_result_verify +
// sgtd_result_inlier_entries issue it costs eight waits and some sixty small copies.  Everything is enqueued behind the
// batch without looking at it — the kernels behind the lists check the batch's overflow flags themselves — the small
// results come back as one packed block, then the inlier pairs' entries.
int sgtd_search_frame(sgtd_handle e, const sgtd_desc_soa *q, int64_t nq, sgtd_frame_search *io) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) { e->err = "not available on a multi-device handle"; return SGTD_ERR_UNSUPPORTED; }
  if (!e || !io || nq < 0 || (nq > 0 && (!q || !q->side || !q->label || !q->frame)) || io->capacity < 0) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  const int cn = e->dc.cand_num;
  const bool lists_only = (io->flags & SGTD_FRAME_LISTS_ONLY) != 0;     // candidate_selector alone: no verification, every pair of every list
  io->n_cand = 0; io->n_inliers = 0;
#ifdef SGTD_EXP_FRAME_LAPS      // host time of the call by part, to stderr (an experiment build)
  struct Laps {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), t = t0;
    std::string s;
    void lap(const char *what) { const auto n = std::chrono::steady_clock::now(); char b[64]; snprintf(b, sizeof b, " %s %.0f", what, std::chrono::duration<double, std::micro>(n - t).count()); s += b; t = n; }
    ~Laps() { fprintf(stderr, "[frame laps us]%s | all %.0f\n", s.c_str(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
  } laps;
#define LAP(x) laps.lap(x)
#else
#define LAP(x) do { } while (0)
#endif
  CHK(settle_tail(e));
  // ---- sgtd_query_descs without its waits
  e->nq = 1;
  e->q_stride = std::max<long long>(nq, 1);
  e->last_kind = 2;
  e->diag = false;
  e->rec_rate_cap = e->stats.overflowed ? e->rec_rate_cap : std::min<u32>(256, e->rec_rate_cap * 2);
  CHK(ensure_store(e, e->qd, (size_t)e->q_stride));
  CHK(ensure(e, e->q_count, sizeof(u32)));
  CHK(copy_in(e, e->qd, 0, (size_t)nq, q, /*wait=*/false));
  LAP("descriptors_in");
  e->thr2_pending = nq;      // the descriptors' sweep records (thresholds, gate masks): launch_select writes them — one frame per call in its one launch
  const u32 cnt = (u32)nq;
  CHK(h2d(e, e->q_count.p, &cnt, sizeof(u32)));      // (staged: the bytes are copied out of `cnt` here)
  if (e->timing) HIPCHK(hipEventRecord(e->ev[EV_START], e->stream));
  const bool deferred = e->defer_lists;
  e->defer_lists = false;
  const int ls = launch_select(e);
  e->defer_lists = deferred;
  CHK(ls);
  LAP("select_launches");
  const int *ovf = reinterpret_cast<const int *>(e->cursors.as<u32>() + 10);
  // the pairs whose entries go to the caller, and their offsets per candidate: the inlier pairs (compacted below), or — lists only —
  // the match lists as they stand (a one-query batch's pairs start at 0 of the pair buffer; pair_off[cn] = their number)
  const u64 *out_pairs = e->pairs.as<u64>();
  const long long *out_off = e->pair_off.as<long long>();
  if (!lists_only) {
    // ---- candidate_verify behind it, sized by what the pair buffer holds
    CHK(verify_enqueue(e, (int64_t)std::min<size_t>(e->pair_cap, 0xFFFFFFF0u), /*guard=*/true));
    // ---- the inlier pairs of every candidate, compacted by one workgroup per candidate, and the entries they name
    CHK(ensure(e, e->inl_counts, (size_t)SGTD_MAX_CAND * sizeof(u32)));
    CHK(ensure(e, e->inl_off, (size_t)(cn + 1) * sizeof(long long)));
    CHK(ensure(e, e->inl_pairs, std::min<size_t>(e->pair_cap, 0xFFFFFFF0u) * sizeof(u64)));
    if (!e->verify_counted) {     // (the packed-f32 form of the vote pass does not count)
      inlier_count_kernel<<<cn, SGTD_INLIER_CAND_THREADS, 0, e->stream>>>(e->v_inlier.as<unsigned char>(), e->pair_off.as<long long>(), e->n_cand.as<int>(), ovf,
                                                                           e->inl_counts.as<u32>());
      HIPCHK(hipGetLastError());
    }
    inlier_compact_kernel<<<cn, SGTD_INLIER_CAND_THREADS, 0, e->stream>>>(e->pairs.as<u64>(), e->v_inlier.as<unsigned char>(), e->pair_off.as<long long>(),
                                                                           e->n_cand.as<int>(), ovf, e->inl_counts.as<u32>(), cn, e->inl_pairs.as<u64>(),
                                                                           e->inl_off.as<long long>());
    HIPCHK(hipGetLastError());
    out_pairs = e->inl_pairs.as<u64>();
    out_off = e->inl_off.as<long long>();
  }
  // the results go where the host reads them: page-locked memory the kernels write over the link — the packed block first (the
  // handle's), then the inlier pairs' entries and query indices: straight into the CALLER's arrays when those are page-locked
  // (sgtd_host_alloc: the 23 MB of a frame's 160 000 pairs on a 10 000-frame map cross the link once, at its rate, and the call
  // has ONE wait), else into the handle's own block, room for `room` pairs, and from there with memcpy.  (Copies device ->
  // host cost 0.15 ms each on this runtime and ran at 16 GB/s: eight of them for the entries were 1.4 ms of the call.)
  const size_t pack_bytes = (frame_pack_bytes(cn) + 24 + 255) & ~(size_t)255;
  const size_t ucap = (size_t)io->capacity;
  const sgtd_desc_soa &o = io->entries;
  static const bool direct_on = [] { const char *v = getenv("SGTD_FRAME_DIRECT"); return !(v && !atoi(v)); }();
  const bool direct = direct_on && ucap > 0 && device_can_write(io->inlier_q_idx, ucap * 4) && device_can_write(o.side, ucap * 24) && device_can_write(o.angle, ucap * 24) &&
                      device_can_write(o.center, ucap * 24) && device_can_write(o.vertex, ucap * 36) && device_can_write(o.label, ucap * 12) &&
                      device_can_write(o.node_id, ucap * 12) && device_can_write(o.frame, ucap * 4);
  DescArrays host_out{};
  int *host_qi = nullptr;
  auto gather = [&](size_t room) -> int {      // entries of the first `room` inlier pairs (the count is on the device)
    const size_t need = pack_bytes + (direct ? 0 : room * 144);
    if (need > e->frame_host_bytes) {
      HIPCHK(hipStreamSynchronize(e->stream));          // (nothing may still be writing the block that is given back)
      if (e->frame_host) (void)hipHostFree(e->frame_host);
      e->frame_host = nullptr; e->frame_host_bytes = 0;
      void *hp = nullptr;
      if (hipHostMalloc(&hp, need, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); e->err = "hipHostMalloc of the frame results failed"; return SGTD_ERR_HIP; }
      e->frame_host = static_cast<char *>(hp); e->frame_host_bytes = need;
    }
    if (direct) {
      DescArrays user{};
      user.side = o.side; user.angle = o.angle; user.center = o.center; user.vertex = o.vertex; user.label = o.label; user.frame = o.frame; user.node_id = o.node_id;
      gather_pair_entries_kernel<<<1024, 256, 0, e->stream>>>(out_pairs, out_off + cn, (long long)ucap, io->inlier_q_idx, e->tab.view(), user, ovf);
      HIPCHK(hipGetLastError());
      return SGTD_OK;
    }
    char *at = e->frame_host + pack_bytes;               // (doubles first: every array stays aligned to its element)
    host_out.side = reinterpret_cast<double *>(at); at += room * 24;
    host_out.angle = reinterpret_cast<double *>(at); at += room * 24;
    host_out.center = reinterpret_cast<double *>(at); at += room * 24;
    host_out.vertex = reinterpret_cast<float *>(at); at += room * 36;
    host_out.label = reinterpret_cast<int *>(at); at += room * 12;
    host_out.node_id = reinterpret_cast<int *>(at); at += room * 12;
    host_out.frame = reinterpret_cast<u32 *>(at); at += room * 4;
    host_qi = reinterpret_cast<int *>(at);
    host_out.qrec = nullptr;
    gather_pair_entries_kernel<<<(unsigned)std::min<long long>(grid_for((long long)room * 8, 256), 1024), 256, 0, e->stream>>>(
        out_pairs, out_off + cn, (long long)room, host_qi, e->tab.view(), host_out, ovf);
    HIPCHK(hipGetLastError());
    return SGTD_OK;
  };
  if (e->frame_inl_cap == 0) e->frame_inl_cap = 16384;
  CHK(gather(direct ? ucap : e->frame_inl_cap));
  pack_frame_kernel<<<1, 256, 0, e->stream>>>(e->cursors.as<u32>(), e->n_cand.as<int>(), e->q_M.as<u32>(), e->q_pair_base.as<u32>(), e->q_count.as<u32>(),
                                              e->q_P.as<unsigned long long>(), e->cand_frame.as<int>(), e->cand_votes.as<int>(),
                                              e->pair_off.as<long long>(), lists_only ? nullptr : e->v_score.as<double>(), lists_only ? nullptr : e->v_pose.as<double>(), out_off,
                                              cn, reinterpret_cast<unsigned char *>(e->frame_host), e->totals.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  LAP("verify_and_result_launches");
  CHK(xfer_sync(e));                                        // ---- the call's one wait (nothing is queued for it: it also gives the staging back)
  struct { const unsigned char *p; const unsigned char *data() const { return p; } } pack{reinterpret_cast<const unsigned char *>(e->frame_host)};
  unsigned long long tot[3];                                // the handle's running totals, as sync_batch reads them
  std::memcpy(tot, pack.data() + frame_pack_bytes(cn), sizeof(tot));
  LAP("wait_1");
  const u32 *w = reinterpret_cast<const u32 *>(pack.data());
  if (w[10] | w[11]) {
    // the batch outgrew a work buffer (a first frame, a frame unlike the ones before): sgtd_sync re-runs it, then the
    // calls this one stands for, one after the other
    CHK(sync_batch(e));
    if (lists_only) {
      std::vector<int64_t> off((size_t)cn + 1, 0);
      CHK(sgtd_result_candidates(e, &io->n_cand, io->cand_frame, io->cand_votes, off.data()));
      if (io->pair_off) std::memcpy(io->pair_off, off.data(), off.size() * sizeof(int64_t));
      if (io->inlier_off) std::memcpy(io->inlier_off, off.data(), off.size() * sizeof(int64_t));
      const int64_t total = off[(size_t)io->n_cand];
      io->n_inliers = total;
      if (total > io->capacity) return SGTD_ERR_CAPACITY;
      if (total == 0) return SGTD_OK;
      std::vector<int64_t> ids((size_t)total);
      std::vector<int32_t> qi;
      int64_t got = 0;
      if (!io->inlier_q_idx) qi.resize((size_t)total);
      CHK(sgtd_result_pairs(e, 0, io->inlier_q_idx ? io->inlier_q_idx : qi.data(), ids.data(), total, &got));
      return sgtd_fetch_entries(e, ids.data(), total, &io->entries);
    }
    CHK(sgtd_verify(e));
    CHK(sgtd_result_candidates(e, &io->n_cand, io->cand_frame, io->cand_votes, io->pair_off));
    CHK(sgtd_result_verify(e, 0, io->score, io->pose));
    std::vector<int64_t> off((size_t)cn + 1);
    const int st = sgtd_result_inlier_entries(e, 0, off.data(), io->inlier_q_idx, &io->entries, io->capacity, &io->n_inliers);
    if (io->inlier_off) std::memcpy(io->inlier_off, off.data(), off.size() * sizeof(int64_t));
    return st;
  }
  // the host-side state every sgtd_result_* call reads, as sync_batch leaves it
  const int *cf = reinterpret_cast<const int *>(pack.data() + 72), *cv = cf + cn;
  const long long *po = reinterpret_cast<const long long *>(pack.data() + 72 + (size_t)cn * 8);
  const double *sc = reinterpret_cast<const double *>(po + cn + 1), *ps = sc + cn;
  const long long *ioff = reinterpret_cast<const long long *>(ps + (size_t)cn * 12);
  if (!(e->h_count.resize(1) && e->h_pair_base.resize(2) && e->h_q_M.resize(1) && e->h_q_P.resize(1) && e->h_n_cand.resize(1) &&
        e->h_cand_frame.resize((size_t)cn) && e->h_cand_votes.resize((size_t)cn) && e->h_pair_off.resize((size_t)cn + 1))) {
    e->err = "hipHostMalloc of the result tables failed";
    return SGTD_ERR_HIP;
  }
  e->h_count[0] = w[15]; e->h_pair_base[0] = 0; e->h_pair_base[1] = w[14]; e->h_q_M[0] = w[13];
  std::memcpy(&e->h_q_P[0], pack.data() + 64, 8);
  e->h_n_cand[0] = (int)w[12];
  std::memcpy(e->h_cand_frame.data(), cf, (size_t)cn * 4); std::memcpy(e->h_cand_votes.data(), cv, (size_t)cn * 4);
  std::memcpy(e->h_pair_off.data(), po, ((size_t)cn + 1) * 8);
  {
    sgtd_stats &st = e->stats;
    unsigned long long swept = 0;
    std::memcpy(&swept, w + 6, 8);
    st.overflowed = 0; st.last_list_moves = w[9];
    st.last_queries = 1; st.last_D = w[15]; st.last_P = (int64_t)e->h_q_P[0]; st.last_M = w[13]; st.last_P_swept = (int64_t)swept;
    st.last_cand_pairs = po[cn];
    if (e->totals.p) { st.batches_total = (int64_t)tot[0]; st.overflow_launches_total = (int64_t)tot[1]; st.list_moves_total = (int64_t)tot[2]; }
  }
  stage_times(e);
  e->batch_synced = true;
  e->verified = !lists_only;
  io->n_cand = (int32_t)w[12];
  if (io->cand_frame) std::memcpy(io->cand_frame, cf, (size_t)cn * 4);
  if (io->cand_votes) std::memcpy(io->cand_votes, cv, (size_t)cn * 4);
  if (io->pair_off) std::memcpy(io->pair_off, po, ((size_t)cn + 1) * 8);
  if (io->score && !lists_only) std::memcpy(io->score, sc, (size_t)cn * 8);
  if (io->pose && !lists_only) std::memcpy(io->pose, ps, (size_t)cn * 96);
  if (io->inlier_off) std::memcpy(io->inlier_off, ioff, ((size_t)cn + 1) * 8);
  const int64_t n_inl = ioff[cn];
  io->n_inliers = n_inl;
  if (n_inl > io->capacity) return SGTD_ERR_CAPACITY;       // (everything else is valid; sgtd_result_inlier_entries with more room gives the pairs)
  if (n_inl == 0) return SGTD_OK;
  if (direct) { LAP("entries_in_place"); return SGTD_OK; }  // (the kernel wrote the caller's arrays)
  if ((size_t)n_inl > e->frame_inl_cap) {                   // more inlier pairs than the gather had room for: once more, with room
    e->frame_inl_cap = (size_t)n_inl + (size_t)n_inl / 2;   // (io, cf .. ioff were copied out above: the block may move)
    CHK(gather(e->frame_inl_cap));
    HIPCHK(hipStreamSynchronize(e->stream));
  } else if ((size_t)n_inl * 2 > e->frame_inl_cap) {
    e->frame_inl_cap = (size_t)n_inl * 2;                   // (room for the next frame; host_out still names this frame's arrays)
  }
  LAP("host_tables");
  const size_t n = (size_t)n_inl;
  if (io->inlier_q_idx) std::memcpy(io->inlier_q_idx, host_qi, n * sizeof(int));
  if (o.side) std::memcpy(o.side, host_out.side, n * 24);
  if (o.angle) std::memcpy(o.angle, host_out.angle, n * 24);
  if (o.center) std::memcpy(o.center, host_out.center, n * 24);
  if (o.vertex) std::memcpy(o.vertex, host_out.vertex, n * 36);
  if (o.label) std::memcpy(o.label, host_out.label, n * 12);
  if (o.node_id) std::memcpy(o.node_id, host_out.node_id, n * 12);
  if (o.frame) std::memcpy(o.frame, host_out.frame, n * 4);
  LAP("entries_out");
  return SGTD_OK;
}
#undef LAP

int sgtd_save_table(sgtd_handle e, const char *path) {
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e || !path) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  FILE *f = fopen(path, "wb");
  if (!f) { e->err = std::string("cannot open for writing: ") + path; return SGTD_ERR_IO; }
  TableHeader h{};
  h.side_resolution = e->cfg.std_side_resolution; h.min_len = e->cfg.descriptor_min_len; h.max_len = e->cfg.descriptor_max_len;
  h.near_num = e->cfg.descriptor_near_num; h.have_frames = e->have_frames ? 1 : 0;
  h.current_frame_id = e->current_frame_id; h.frame_lo = e->frame_lo; h.frame_hi = e->frame_hi;
  h.n_entries = e->n_entries; h.n_add_calls = e->n_add_calls;
  int st = (fwrite(kTableMagic, 1, 8, f) == 8 && fwrite(&h, sizeof(h), 1, f) == 1) ? SGTD_OK : SGTD_ERR_IO;
  if (st == SGTD_OK) st = stream_table(e, f, (size_t)e->n_entries, true);
  if (fclose(f) != 0 && st == SGTD_OK) st = SGTD_ERR_IO;
  if (st == SGTD_ERR_IO) e->err = std::string("write failed: ") + path;
  return st;
}

int sgtd_load_table(sgtd_handle e, const char *path) {
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e || !path) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (e->attached_to) { e->err = "the table belongs to another handle (sgtd_attach_table): load into its owner"; return SGTD_ERR_STATE; }
  CHK(settle_pending(e));
  e->table_version++;
  FILE *f = fopen(path, "rb");
  if (!f) { e->err = std::string("Error opening file: ") + path; return SGTD_ERR_IO; }
  TableHeader h{};
  {
    const int hs = read_table_header(f, path, e->cfg, h, e->err);
    if (hs != SGTD_OK) { fclose(f); return hs; }
  }
  int st = ensure_store(e, e->tab, (size_t)std::max<int64_t>(h.n_entries, 1), false);
  if (st == SGTD_OK) st = stream_table(e, f, (size_t)h.n_entries, false);
  fclose(f);
  if (st == SGTD_OK && h.n_entries > 0) {
    // the votes are indexed by frame - frame_lo: every stored frame id must lie in the header's range
    int bad = 0;
    st = ensure(e, e->bad_flag, sizeof(int));
    if (st == SGTD_OK) {
      HIPCHK(hipMemsetAsync(e->bad_flag.p, 0, sizeof(int), e->stream));
      frame_range_check_kernel<<<grid_for(h.n_entries, 256), 256, 0, e->stream>>>(e->tab.frame.as<u32>(), h.n_entries, h.frame_lo, h.frame_hi,
                                                                                   e->bad_flag.as<int>());
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&bad, e->bad_flag.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      if (bad) st = SGTD_ERR_IO;
    }
  }
  if (st != SGTD_OK) {
    if (st == SGTD_ERR_IO) e->err = std::string(path) + ": truncated or damaged table file";
    e->n_entries = 0; e->have_frames = false; e->finalized = false; e->batch_valid = false;
    e->seg[0].built = false; e->seg[1].built = false;
    return st;
  }
  e->n_entries = h.n_entries; e->n_add_calls = h.n_add_calls; e->current_frame_id = h.current_frame_id;
  e->have_frames = h.have_frames != 0; e->frame_lo = h.frame_lo; e->frame_hi = h.frame_hi;
  e->seg[0].built = false; e->seg[1].built = false;
  e->finalized = false;
  e->batch_valid = false;
  return SGTD_OK;
}

int sgtd_host_alloc(size_t bytes, void **out) {
  if (!out) return SGTD_ERR_INVALID;
  *out = nullptr;
  if (bytes == 0) return SGTD_OK;
  return hipHostMalloc(out, bytes, hipHostMallocPortable) == hipSuccess ? SGTD_OK : SGTD_ERR_HIP;
}

int sgtd_host_free(void *p) {
  if (!p) return SGTD_OK;
  return hipHostFree(p) == hipSuccess ? SGTD_OK : SGTD_ERR_INVALID;
}

int sgtd_fetch_entries(sgtd_handle e, const int64_t *db_entry, int64_t n, sgtd_desc_soa *out) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) return multi::fetch_entries(e, db_entry, n, out);
  if (!e || n < 0 || (n > 0 && (!db_entry || !out))) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  CHK(view_current(e));
  if (n == 0) return SGTD_OK;
  bool contiguous = true;
  for (int64_t i = 0; i < n; i++) {
    if (db_entry[i] < 0 || db_entry[i] >= e->n_entries) return SGTD_ERR_INVALID;
    if (i && db_entry[i] != db_entry[i - 1] + 1) contiguous = false;
  }
  if (contiguous) return copy_out(e, e->tab, (size_t)db_entry[0], (size_t)n, out, 0);
  // scattered ids (a match list): gathered on the device, then one copy per field
  CHK(ensure(e, e->fetch_idx, (size_t)n * sizeof(long long)));
  CHK(ensure_store(e, e->fetch, (size_t)n));
  CHK(h2d(e, e->fetch_idx.p, db_entry, (size_t)n * sizeof(long long)));
  gather_entries_kernel<<<grid_for(n, 256), 256, 0, e->stream>>>(e->fetch_idx.as<long long>(), n, e->tab.view(), e->fetch.view());
  HIPCHK(hipGetLastError());
  return copy_out(e, e->fetch, 0, (size_t)n, out, 0);
}

int sgtd_table_dump(sgtd_handle e, int64_t *keys, int64_t *bucket_off, int64_t *entry_ids,
                    int64_t cap_buckets, int64_t cap_entries) {
  PinScope pin_scope(e && !e->grp ? e : nullptr);
  if (e && e->grp) { e->err = "not available on a multi-device handle (use the per-device form, sgtd_amd/dist.py)"; return SGTD_ERR_UNSUPPORTED; }
  if (!e) return SGTD_ERR_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device_id));
  if (e->attached_to && e->n_seg > 1) { e->err = "dump the table through its owner"; return SGTD_ERR_STATE; }
  CHK(do_finalize(e, /*force_merge=*/true));   // one segment: the dump shows whole buckets
  const sgtd_engine::Segment &S = e->seg[0];
  const int64_t U = S.n_buckets, E = e->n_entries;
  if (U > cap_buckets || E > cap_entries) return SGTD_ERR_CAPACITY;
  if (E == 0) { if (bucket_off) bucket_off[0] = 0; return SGTD_OK; }
  std::vector<u64> k(U);
  std::vector<u32> st(U), pm(E);
  HIPCHK(hipMemcpy(k.data(), S.bucket_key.p, U * sizeof(u64), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(st.data(), S.bucket_start.p, U * sizeof(u32), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(pm.data(), S.perm.p, E * sizeof(u32), hipMemcpyDeviceToHost));
  for (int64_t u = 0; u < U; u++) {
    if (keys) {
      keys[u * 4 + 0] = (int64_t)((k[u] >> 32) & 0xFFFF);
      keys[u * 4 + 1] = (int64_t)((k[u] >> 16) & 0xFFFF);
      keys[u * 4 + 2] = (int64_t)(k[u] & 0xFFFF);
      keys[u * 4 + 3] = (int64_t)(k[u] >> 48);
    }
    if (bucket_off) bucket_off[u] = st[u];
  }
  if (bucket_off) bucket_off[U] = E;
  if (entry_ids) {
    // the probe layout keeps every bucket partitioned by slice; the dump shows each bucket in
    // insertion order (ascending entry id), the order of the reference's bucket vector
    for (int64_t u = 0; u < U; u++) {
      const int64_t lo = st[u], hi = (u + 1 < U) ? (int64_t)st[u + 1] : E;
      std::sort(pm.begin() + lo, pm.begin() + hi);
    }
    for (int64_t p = 0; p < E; p++) entry_ids[p] = pm[p];
  }
  return SGTD_OK;
}

int sgtd_get_stats(sgtd_handle e, sgtd_stats *out) {
  if (e && e->grp) return multi::get_stats(e, out);
  if (!e || !out) return SGTD_ERR_INVALID;
  e->stats.n_entries = e->n_entries;
  e->stats.n_buckets = e->seg[0].n_buckets + (e->n_seg > 1 ? e->seg[1].n_buckets : 0);
  e->stats.n_frames = e->n_add_calls;
  e->stats.hbm_bytes_table = e->n_entries * (int64_t)SGTD_HOT_BYTES;
  e->stats.bucket_len_sq_over_E = e->n_entries > 0 ? (e->seg[0].sum_len_sq + (e->n_seg > 1 ? e->seg[1].sum_len_sq : 0.0)) / (double)e->n_entries : 0.0;
  e->stats.ms_finalize = e->ms_finalize;
  e->stats.tail_entries = e->n_seg > 1 ? e->seg[1].g1 - e->seg[1].g0 : 0;
  *out = e->stats;
  return SGTD_OK;
}

}  // extern "C"

#include "graph_ingest_abi.h"
#include "multi_impl.hip.h"
