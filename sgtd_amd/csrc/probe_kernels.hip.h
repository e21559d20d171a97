// probe_kernels.hip.h — candidate_selector (src/sgtd/src/STDesc.cpp:318-460)
//
//   probe     (:351-400) one wavefront per pair of query descriptors of one home
//             cell: the 27 cells (truncating (int)(side+inc), gate ||side-centre||
//             < 1.5, hash lookup key -> bucket) become one concatenated visit list
//             of non-empty ranges — per cell and half of the second side's
//             interval the thirds of the third side's that a descriptor's threshold
//             box reaches, and the bucket's overflow slice (common.hip.h,
//             table_kernels.hip.h) — that all 64 lanes stream from the 16-B/entry
//             probe layout (one 16-B load per lane and 64 entries, 4 x 64 entries
//             in flight).  The distance test runs in f32 against two squared
//             thresholds that make it conservative on both sides (common.hip.h
//             f32_bounds); the few entries between them are decided on the exact
//             f64 sides with the exact squared threshold (sq_threshold) — every
//             decision is the reference's.  Matches are compacted in visit order
//             by ballot/popcount prefix into a per-descriptor list of 4-byte entry
//             ids (frame and entry in one word, IdMap); restricted to any one map
//             frame that order is the reference's (cell, j) order.
//             Schedule (probe_sorted_kernel): the batch's descriptors are
//             radix-sorted by home cell, descriptors of one home cell share ONE
//             set of 27 bucket lookups (GroupRow), per-XCD ticket queues hand
//             neighbouring cells to waves of one XCD so that the buckets stay in
//             that XCD's L2; votes_kernel counts votes from the lists afterwards.
//   topk      (:423-433) candidate_num rounds of arg-max over the votes:
//             votes desc, frame id asc, stop below 5 votes
//   assemble  (:434-449) one wavefront per 128-descriptor block walks the
//             block's match lists as one flattened stream: pass 1
//             (block_count_kernel) counts matches per candidate slot and
//             compacts the candidate matches, in list order, into a dense
//             per-block list; a scan over blocks gives every (block, slot) its
//             output range; pass 2 (block_write_kernel) splits the dense list
//             by slot — a stable split: equal-slot lanes find each other through
//             commutative LDS ORs, running positions in lane registers — and
//             writes each candidate's match_list_ in the reference's (i, cell, j)
//             order through per-slot 128-B staging lines
//
// Diagnostics (never in the shipped build): -DSGTD_EXP_PHASE adds in-kernel phase
// clocks and counters printed by the host after a few launches; -DSGTD_EXP_TRACE
// records per-wave start/end times of the sweep.
#pragma once
#include "common.hip.h"
#include <type_traits>

struct TableView {
  const HotEntry *ent;         // [E] probe order: by key, by slice inside a bucket, insertion order inside a slice
  IdMap map;                   // entry id -> frame, insertion index
  const double *cold_side;     // [E*3] exact sides in insertion order (undecided f32 tests, diagnostic build)
  const BucketDir *dir;        // [U] bucket directory
  const HashSlot *hash;
  u32 hash_mask;
  u32 n_entries;
  u32 frame_lo;                // votes are indexed by frame - frame_lo
  u32 frame_span;              // number of vote bins per query
  u32 coarse_at;               // a visit list of more ranges than this is planned without slice pruning (62; SGTD_COARSE_AT: test hook)
};

struct QueryView {
  const double *side;   // [n_slots*3]
  const QueryRec *qrec; // [n_slots] sweep record of the descriptor (thresholds, gate mask)
  const int *label;     // [n_slots*3]
  const u32 *frame;     // [n_slots]
  const u32 *count;     // [n_queries] descriptors per query
  long long stride;     // descriptor slots per query
  int n_queries;
};

struct ProbeBuffers {
  u32 *rec;             // [rec_cap] match records: the entry's id (frame and entry in one word, common.hip.h IdMap)
  unsigned char *rec_cell;  // [rec_cap] voxel_round index (diagnostic build only)
  double *rec_dis;      // [rec_cap] distance (diagnostic build only)
  u32 rec_cap;
  unsigned long long *rec_cursor;   // global slab cursor (64-bit: requests can add up beyond 2^32)
  unsigned long long *rec_need;     // matches that found no room (sizes the regrown buffer)
  unsigned long long *swept;        // table entries the sweep really loaded (after slice pruning)
  // per table segment sg and descriptor slot d, at [sg * seg_stride + d] (the sweep of segment sg
  // gets the three pointers advanced to its part):
  u32 *list_ptr;        // first record of the descriptor's list from that segment
  u32 *n_visit;         // entries the reference's loop visits for the descriptor there (STDesc.cpp:372)
  u32 *n_match;         // matches of the descriptor there
  long long seg_stride; // descriptor slots of the batch
  int n_seg;            // table segments swept (main, tail)
  u32 *votes;           // [n_queries * frame_span]
  int *overflow;        // [2]: 0 match records, 1 candidate pairs
  u32 id_bits;          // a record's local frame (frame - table frame_lo) is rec >> id_bits
  // records whose f32 test fell between the two thresholds: stored provisionally as matches,
  // queued here and decided on the exact sides by resolve_undecided_kernel right after the sweep
  uint2 *amb_queue;     // [amb_cap] (record index, descriptor slot)
  u32 *amb_count;
  u32 amb_cap;
};

#define SGTD_PROBE_THREADS 256
#define SGTD_PROBE_CHUNK 128    // query descriptors per assemble block
#define SGTD_REC_SLAB 8192u     // match records a wave takes from the global cursor at once
#define SGTD_SUB_DESCS 32       // descriptors per prefix sub-block inside an assemble block
#ifndef SGTD_VOTE_WORDS
#define SGTD_VOTE_WORDS 2        // 64-quad words of records in flight per wave in the vote pass
#endif
#ifndef SGTD_WRITE_WORDS
#define SGTD_WRITE_WORDS 2       // 64-pair words of the compact list in flight per wave in block_write
#endif
#ifndef SGTD_PROBE_UNROLL
#define SGTD_PROBE_UNROLL 4     // 64-entry words whose loads are in flight together
#endif

// per-descriptor results of the sweep (stored once per ticket by the caller)
struct DescResult {
  u32 ptr, visit, match;
};

// Loads that were issued before a descriptor's sweep and are first used after it (the next
// descriptor's GroupRow, the next ticket's records): the sweep "touches" them once its first
// load group has returned — vector loads return in order, so they are complete by then —
// otherwise the compiler, which cannot count the stores of the loops in between, would wait
// for every outstanding store (vmcnt(0)) at their first use.
struct PendingLoads {
  uint4 *row = nullptr;
  uint4 *rec = nullptr;
  __device__ __forceinline__ void touch() const {
    if (row) asm volatile("" : "+v"(row->x), "+v"(row->y), "+v"(row->z), "+v"(row->w));
    if (rec) asm volatile("" : "+v"(rec->x), "+v"(rec->y), "+v"(rec->z), "+v"(rec->w));
  }
};

#ifndef SGTD_PAIR
// descriptors of one home cell swept together by a wave (shared plan, locate and loads): 1 or 2.
// (4 at a time halves the loads again but needs 129 VGPRs — 3 waves per SIMD: measured slower)
#define SGTD_PAIR 2
#endif
static_assert(SGTD_PAIR == 1 || SGTD_PAIR == 2, "the pair sweep computes both distances with packed f32 math");

struct WaveSlab {
  u32 next[SGTD_PAIR], end[SGTD_PAIR];   // this wave's private ranges of match records: one bump stream per
                                         // descriptor of a pair, so that every descriptor's list stays contiguous
  u64 swept;                             // entries this wave loaded
#ifdef SGTD_EXP_PHASE
  u64 ph[8];
#endif
};
#ifdef SGTD_EXP_PHASE
__device__ unsigned long long g_phase[8];
__device__ unsigned long long g_words[2];   // 64-entry words swept, load groups (trips) issued
#define PH_T() __builtin_readcyclecounter()
#define PH_ADD(i, t0) do { const u64 _n = PH_T(); slab.ph[i] += _n - (t0); (t0) = _n; } while (0)
#else
#define PH_T() 0ull
#define PH_ADD(i, t0) do { } while (0)
#endif

// what the sweep needs about K query descriptors of ONE home cell (they share the GroupRow); the
// loads are issued one step ahead, the per-lane plan is derived right before the sweep
template <int K>
struct DescSet {
  uint4 row;            // lane l < 54: 16-B quarter l of the group's 27 directory rows
  double q0[K], q1[K], q2[K], thr2[K];
  float lo2[K], hi2[K]; // conservative f32 thresholds (f32_bounds)
  float t_up[K];        // f32 upper bound of the match threshold
  u32 qframe[K];
  u32 gate[K];          // the descriptor's 27-bit gate mask
  u32 slot[K];          // descriptor slot d
};

// The visit list as the sweep walks it: the non-empty ranges of table entries, in the order of
// the reference's cells (inside a cell: lower half, upper half, overflow slice), numbered
// j = 0..n-1 and held one per lane: offc = exclusive offset of range j in the list,
// dlc = start - offc, its cell (diagnostic sweep) and, for a pair, per descriptor the penalty
// 0 / +inf that starts the squared-distance sum (+inf: the cell fails that descriptor's gate).
// For a pair the list is the union of the two descriptors' lists.  Lane n stands for one more
// range that starts at `total` and maps onto the sentinel entries behind the table's last entry
// (sides +inf: no match), so the lanes of the last 64-entry word beyond the list need no special
// case; lanes beyond n hold offc = 0xFFFFFFFF.
template <int K>
struct DescPlan {
  u32 offc;
  u32 dlc, cellc;
  float penc[K];
  u32 n, total;         // wave-uniform
  u32 ref_visits[K];    // wave-uniform: entries the reference's loop visits (all slices of the gated cells)
};

// ---------------------------------------------------------------------------
// sweep: the batch's descriptors are visited in the order of a locality key
// (label code, cell x, y, z) and each XCD walks one contiguous eighth of that
// order, so the buckets a wave needs were just used by its neighbours on the
// same XCD and come from that XCD's L2 instead of HBM.  Results are independent
// of the order: every descriptor writes its own list.
// ---------------------------------------------------------------------------
// Home key of a descriptor slot = (label code, (int)side0, (int)side1, (int)side2) packed
// with `cbits` bits per cell coordinate; invalid slots sort last.  The 27 probed cells
// (int)(side+inc) are a function of the home cell alone (per axis n = (int)s gives
// {n ? n-1 : 0, n, n+1}), so all descriptors of the batch with equal home key share ONE set
// of bucket lookups (a GroupRow); gate and slice ranges are per descriptor.
// exclusive prefix of the per-query descriptor counts and their total (one workgroup)
__global__ __launch_bounds__(256) void query_prefix_kernel(const u32 *count, u32 *q_prefix, int n_queries,
                                                           u32 *n_valid) {
  __shared__ u32 lds[256 / SGTD_WAVE + 1];
  u32 carry = 0;
  for (int q0 = 0; q0 < n_queries; q0 += 256) {
    const int q = q0 + threadIdx.x;
    const u32 v = (q < n_queries) ? count[q] : 0;
    u32 tot;
    const u32 ex = block_excl_scan(v, lds, tot);
    if (q < n_queries) q_prefix[q] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *n_valid = carry;
}

// keys/vals are compact: the descriptor (q, i) goes to index q_prefix[q] + i
__global__ void home_keys_kernel(QueryView Q, const u32 *q_prefix, u64 *keys, u32 *vals, long long n_slots,
                                 int cbits) {
  const long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_slots) return;
  const int q = (int)(d / Q.stride);
  const u32 i = (u32)(d - (long long)q * Q.stride);
  if (i >= Q.count[q]) return;
  const u64 code = label_code(Q.label[d * 3 + 0], Q.label[d * 3 + 1], Q.label[d * 3 + 2]);
  const u64 cmask = (1ull << cbits) - 1ull;
  // a coordinate that does not fit cbits-1 bits becomes the all-ones marker: such a
  // descriptor never shares a group (group_heads_kernel), so aliasing cannot merge cells
  const u64 x = min((u64)(u32)(int)Q.side[d * 3 + 0], cmask), y = min((u64)(u32)(int)Q.side[d * 3 + 1], cmask),
            z = min((u64)(u32)(int)Q.side[d * 3 + 2], cmask);
  const u32 idx = q_prefix[q] + i;
  keys[idx] = (((code << cbits | x) << cbits | y) << cbits) | z;
  vals[idx] = (u32)d;
}

// group id of every sorted position from the SORTED home keys: flags[p] = 1 where the key
// changes or where the next key carries an overflow marker (such a descriptor is a group of
// its own: its key does not name its cell).  flags[p] says whether position p+1 starts a new
// group, so the EXCLUSIVE scan of the flags is the group id of p (position 0 is group 0).
__global__ void group_heads_kernel(const u64 *keys, const u32 *n_valid_p, u32 *flags, long long n, int cbits) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  u32 head = 0;
  if (p + 1 < (long long)*n_valid_p) {
    const u64 a = keys[p + 1], b = keys[p];
    const u64 cmask = (1ull << cbits) - 1ull;
    head = (a != b) || ((a & cmask) == cmask) || (((a >> cbits) & cmask) == cmask) ||
           (((a >> (2 * cbits)) & cmask) == cmask);
  }
  flags[p] = head;
}

// sorted records + first sorted position of every group + the number of groups; four
// threads per position, one 16-B quarter of the record each (coalesced 64-B reads, 1-KB writes)
__global__ void sorted_desc_kernel(QueryView Q, const u32 *order, const u32 *gid, const u32 *n_valid_p,
                                   QueryRec *out, u32 *group_first, u32 *n_groups, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long p = t >> 2;
  const int qtr = (int)(t & 3);
  const long long nv = (long long)*n_valid_p;
  if (p >= n || p >= nv) return;
  const u32 g = gid[p];
  const u32 d = order[p];
  uint4 v = reinterpret_cast<const uint4 *>(Q.qrec + d)[qtr];
  if (qtr == 2) {
    v.z = g; v.w = d;
    if (p == 0 || gid[p - 1] != g) group_first[g] = (u32)p;
    if (p == nv - 1) *n_groups = g + 1;
  }
  reinterpret_cast<uint4 *>(out + p)[qtr] = v;
}

// One GroupRow per home cell of the batch: the 27 ungated bucket lookups (STDesc.cpp:358-371
// minus the gate), each answered with the bucket's directory row {start, cum[5]} (all zero if
// the table has no such bucket): 27 x 32 B at rows + g * SGTD_GROUP_ROW_BYTES.
// 32 lanes per group, grid-stride.
__global__ __launch_bounds__(256) void group_resolve_kernel(TableView T, QueryView Q, const u32 *order,
                                                            const u32 *group_first, const u32 *n_groups_p,
                                                            const u32 *n_valid_p, unsigned char *rows) {
  const int c = (int)(threadIdx.x & 31);
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 5;
  const long long n_groups = (*n_valid_p) ? (long long)*n_groups_p : 0;
  for (long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5; g < n_groups; g += stride) {
    if (c >= SGTD_NCELL) continue;
    const long long d = (long long)order[group_first[g]];
    uint4 lo = make_uint4(0, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
    const double q0 = Q.side[d * 3 + 0], q1 = Q.side[d * 3 + 1], q2 = Q.side[d * 3 + 2];
    const u32 code = label_code(Q.label[d * 3 + 0], Q.label[d * 3 + 1], Q.label[d * 3 + 2]);
    const int ix = c / 9 - 1, iy = (c / 3) % 3 - 1, iz = c % 3 - 1;  // voxel_round order (:327-333)
    // (int)(side + inc) from the HOME cell n = (int)side, the same for every member of the group:
    // n + inc, and 0 for inc = -1 at n = 0 (C truncation of a value in (-1, 0)).  The float form
    // differs only for a side that is the largest double below a power of two (side + 1 rounds up
    // across the integer): such a descriptor's own gate mask, computed with the float form like
    // the reference, excludes that cell on both sides.
    // (a descriptor with a negative side — outside the envelope — is a group of its own,
    // home_keys_kernel, and keeps the float form)
    const bool nonneg = q0 >= 0.0 && q1 >= 0.0 && q2 >= 0.0;
    const int x = nonneg ? max((int)q0 + ix, 0) : (int)(q0 + (double)ix);
    const int y = nonneg ? max((int)q1 + iy, 0) : (int)(q1 + (double)iy);
    const int z = nonneg ? max((int)q2 + iz, 0) : (int)(q2 + (double)iz);
    if (x >= 0 && y >= 0 && z >= 0 && x < 65536 && y < 65536 && z < 65536) {
      const u64 key = pack_key(code, (u32)x, (u32)y, (u32)z);
      u32 h = hash_key(key) & T.hash_mask;
      while (true) {
        const HashSlot s = T.hash[h];
        if (s.key == key) {
          const uint4 *row = reinterpret_cast<const uint4 *>(T.dir + s.bucket);
          lo = row[0]; hi = row[1];
          break;
        }
        if (s.key == SGTD_EMPTY_KEY) break;
        h = (h + 1) & T.hash_mask;
      }
    }
    uint4 *out = reinterpret_cast<uint4 *>(rows + (size_t)g * SGTD_GROUP_ROW_BYTES) + 2 * c;
    out[0] = lo;     // start, cum[0..2]
    out[1] = hi;     // cum[3..6]
  }
}

// GroupRow + the descriptors' gate masks (:366-369) + their thresholds -> the visit list.
// Lane 2 c + h holds one 16-B half of cell c's directory row: even lanes {start, cum0, cum1,
// cum2}, odd lanes {cum3, cum4, cum5, cum6}, and stands for half h of the cell: the thirds of
// that half which hold entries with |side1 - q1| <= thr' and |side2 - q2| <= thr' — an entry
// outside cannot match: its squared distance, as the reference computes it, is at least
// fl(d * d) >= thr2 for that axis.  The odd lanes also stand for the cell's overflow slice
// (all of it; most buckets have none).  The non-empty ranges are compacted, in cell order, by
// two forward permutes (a lane nobody writes receives 0: the two results are OR-ed), and the
// offsets are the scan of the compacted lengths.
// K = 2: the union of the two descriptors' ranges; the sweep tests every loaded entry against
// both and starts each sum from the descriptor's own gate penalty (a slice too many is harmless,
// a cell too many is not).
__device__ __forceinline__ void reached_slices(float dq, float t_up, int n, int &lo, int &hi) {
  // slices of the cell interval reached by [q - t, q + t]; slice s holds the entries with
  // (side + 0.5 - cell) * n in [s, s + 1).  q - cell is below 2.5 in magnitude for every gated
  // cell: the bounds are computed in f32 from it (error < 1e-6) with a margin of 1e-4 slice on
  // top of the threshold's own upward rounding; visiting a slice too many is harmless.
  const float a = ((dq - t_up) + 0.5f) * (float)n - 1e-4f;
  const float b = ((dq + t_up) + 0.5f) * (float)n + 1e-4f;
  lo = !(a > 0.0f) ? 0 : (a >= (float)n ? n : (int)a);          // floor, clamped to [0, n]; NaN -> 0
  hi = b < 0.0f ? -1 : (!(b < (float)n) ? n - 1 : (int)b);      // floor, clamped to [-1, n - 1]; NaN -> n - 1
}

template <int K>
__device__ __forceinline__ DescPlan<K> plan_from_group_row(const DescSet<K> &f, u32 n_entries, u32 coarse_at) {
  const int lane = lane_id();
  const int c = lane >> 1;
  const bool odd = lane & 1;
  const int half = lane & 1;
  // neighbour lane of the pair (quad_perm [1,0,3,2]): the other half of the cell's row
  const u32 px = (u32)__builtin_amdgcn_update_dpp(0, (int)f.row.x, 0xB1, 0xf, 0xf, false);
  const u32 pz = (u32)__builtin_amdgcn_update_dpp(0, (int)f.row.z, 0xB1, 0xf, 0xf, false);
  const u32 pw = (u32)__builtin_amdgcn_update_dpp(0, (int)f.row.w, 0xB1, 0xf, 0xf, false);
  // cumulative counts around this lane's half: before its first third, after each third
  const u32 cm1 = odd ? pw : 0u, c0 = odd ? f.row.x : f.row.y, c1 = odd ? f.row.y : f.row.z, c2 = odd ? f.row.z : f.row.w;
  const u32 bstart = odd ? px : f.row.x;
  const int iy = (c / 3) % 3 - 1, iz = c % 3 - 1;
  int s_lo = SGTD_ZSLICES, s_hi = -1;
  bool live = false;
  DescPlan<K> pl;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const bool lv = lane < SGTD_NRANGE && ((f.gate[k] >> c) & 1u);
    int y_lo, y_hi, z_lo, z_hi;
    reached_slices((float)(f.q1[k] - (double)(int)(f.q1[k] + (double)iy)), f.t_up[k], SGTD_YSLICES, y_lo, y_hi);
    reached_slices((float)(f.q2[k] - (double)(int)(f.q2[k] + (double)iz)), f.t_up[k], SGTD_ZSLICES, z_lo, z_hi);
    if (lv && half >= y_lo && half <= y_hi && z_hi >= z_lo) { s_lo = min(s_lo, z_lo); s_hi = max(s_hi, z_hi); }
    live |= lv;
    // the reference's loop visits every entry of every gated cell: cum6 of the odd lanes
    pl.ref_visits[k] = wave_sum((lv && odd) ? f.row.w : 0u);
  }
  const u32 before = s_lo == 0 ? cm1 : (s_lo == 1 ? c0 : c1);
  const u32 upto = s_hi == 0 ? c0 : (s_hi == 1 ? c1 : c2);
  u32 start_a = bstart + before;
  u32 len_a = (s_hi >= s_lo) ? upto - before : 0u;            // (s_hi >= s_lo only for a live lane)
  const u32 start_b = px + f.row.z;                           // odd lanes: bucket start + cum5
  const u32 len_b = (odd && live) ? f.row.w - f.row.z : 0u;   // cum6 - cum5
  u64 keep_a = __builtin_amdgcn_ballot_w64(len_a != 0u);
  const u64 keep_b = __builtin_amdgcn_ballot_w64(len_b != 0u);
  if ((u32)(__builtin_popcountll(keep_a) + __builtin_popcountll(keep_b)) > coarse_at) {
    // more ranges than lanes (only with many overflow slices): the halves of a cell as ONE
    // unpruned range, all six sub-cells — a superset of what the descriptors reach
    start_a = f.row.x;
    len_a = (!odd && live) ? pz : 0u;                         // cum5 of the even lane's cell
    keep_a = __builtin_amdgcn_ballot_w64(len_a != 0u);
  }
  const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(keep_a >> 32), __builtin_amdgcn_mbcnt_lo((u32)keep_a, 0u)) +
                    __builtin_amdgcn_mbcnt_hi((u32)(keep_b >> 32), __builtin_amdgcn_mbcnt_lo((u32)keep_b, 0u));
  const u32 n = (u32)(__builtin_popcountll(keep_a) + __builtin_popcountll(keep_b));
  // forward permutes: range -> its number; an empty one writes lane 63, which no number reaches
  const u32 dst_a = (len_a != 0u ? below : 63u) << 2;
  u32 startc = (u32)__builtin_amdgcn_ds_permute((int)dst_a, (int)start_a);
  u32 lenc = (u32)__builtin_amdgcn_ds_permute((int)dst_a, (int)len_a);
  u32 cellc = (u32)__builtin_amdgcn_ds_permute((int)dst_a, c);
  u32 penb[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const bool open = lane < SGTD_NRANGE && ((f.gate[k] >> c) & 1u);
    penb[k] = open ? 0u : 0x7F800000u;
  }
  u32 pencu[K];
#pragma unroll
  for (int k = 0; k < K; k++) pencu[k] = (u32)__builtin_amdgcn_ds_permute((int)dst_a, (int)penb[k]);
  if (keep_b) {     // the overflow slice of cell c follows both of its halves
    const u32 dst_b = (len_b != 0u ? below + (u32)((keep_a >> lane) & 1ull) : 63u) << 2;
    startc |= (u32)__builtin_amdgcn_ds_permute((int)dst_b, (int)start_b);
    lenc |= (u32)__builtin_amdgcn_ds_permute((int)dst_b, (int)len_b);
    cellc |= (u32)__builtin_amdgcn_ds_permute((int)dst_b, c);
#pragma unroll
    for (int k = 0; k < K; k++) pencu[k] |= (u32)__builtin_amdgcn_ds_permute((int)dst_b, (int)penb[k]);
  }
  lenc = (u32)lane < n ? lenc : 0u;
  const u32 inc = wave_incl_scan(lenc);
  const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, SGTD_WAVE - 1);
  pl.n = n;
  pl.total = total;
  pl.offc = (u32)lane < n ? inc - lenc : ((u32)lane == n ? total : 0xFFFFFFFFu);
  pl.dlc = (u32)lane < n ? startc - pl.offc : n_entries - total;       // lane n: the sentinel entries
  pl.cellc = cellc;
#pragma unroll
  for (int k = 0; k < K; k++) pl.penc[k] = __uint_as_float((u32)lane < n ? pencu[k] : 0x7F800000u);
  return pl;
}

// STDesc.cpp:372-399 for K query descriptors of one home cell by one wavefront: streams the
// (union) visit list once, tests every entry against each descriptor, compacts each
// descriptor's matches in visit order into its own list.
// WIDE = false: the probe layout is below 4 GB, so entry addresses are a uniform base + a
// 32-bit byte offset (no quarter-rate 64-bit VALU address arithmetic per entry); the host
// picks the variant from the table size.  Record addresses are a wave-uniform list base + a
// 32-bit lane offset either way.
template <bool DIAG, bool WIDE, int K>
__device__ __forceinline__ void sweep_descriptors(const TableView &T, const ProbeBuffers &B, double rough,
                                                  const DescSet<K> &f, const DescPlan<K> &pl, u64 *bits,
                                                  WaveSlab &slab, DescResult (&result)[K], PendingLoads pending) {
  static_assert(!DIAG || K == 1, "the diagnostic sweep takes one descriptor at a time");
  const int lane = lane_id();
  // per-descriptor constants of the test (wave-uniform)
  float q0f[K], q1f[K], q2f[K], lo2[K], hi2[K];
  u32 qframe[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    q0f[k] = (float)f.q0[k]; q1f[k] = (float)f.q1[k]; q2f[k] = (float)f.q2[k];
    lo2[k] = f.lo2[k]; hi2[k] = f.hi2[k];
    // the query's frame as the ids name it; a frame the table does not hold equals no entry's
    const u32 ql = f.qframe[k] - T.map.frame_lo;
    qframe[k] = ql < T.frame_span ? ql : 0xFFFFFFFFu;
  }
  const u32 id_bits = T.map.bits;
  const double thr = DIAG ? norm3(f.q0[0], f.q1[0], f.q2[0]) * rough : 0.0;   // :356-357
  const u32 total = pl.total;
  u64 ph_t = PH_T(); (void)ph_t;
  // records of one descriptor are contiguous: make sure its stream's slab can take the worst
  // case (every visited entry matches)
  bool fits = true;
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (total && (u64)slab.next[k] + total > (u64)slab.end[k]) {
      // a slab must have room for the worst case of a descriptor (every visit matches) when
      // the descriptor starts, but only the matches stay: slabs of 8 worst cases keep the space
      // abandoned at a slab's end to about an eighth however long the visit lists are
      const u32 take = total > (1u << 28) ? total : max(SGTD_REC_SLAB, 8u * total);
      u64 got = 0;
      if (lane == 0) got = atomicAdd(B.rec_cursor, (unsigned long long)take);
      got = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(got >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)got);
      if (got + take <= (u64)B.rec_cap) { slab.next[k] = (u32)got; slab.end[k] = (u32)got + take; }
      else { slab.next[k] = 0; slab.end[k] = 0; }      // the buffer is exhausted: nothing of this stream fits any more
    }
    fits = fits && ((u64)slab.next[k] + total <= (u64)slab.end[k]);
  }
  if (!fits && lane == 0) B.overflow[0] = 1;
  __builtin_amdgcn_wave_barrier();
  PH_ADD(0, ph_t);

  u32 matches[K];
  u32 *list[K];   // wave-uniform
#pragma unroll
  for (int k = 0; k < K; k++) {
    matches[k] = 0;
    list[k] = B.rec + slab.next[k];
  }
  const u32 n_words = (total + 63u) >> 6;
  slab.swept += total;
#ifdef SGTD_EXP_PHASE
  if (lane == 0 && n_words) {
    atomicAdd(&g_words[0], (unsigned long long)n_words);
    atomicAdd(&g_words[1], (unsigned long long)((n_words + SGTD_PROBE_UNROLL - 1) / SGTD_PROBE_UNROLL));
  }
#endif
  // position -> range.  The starts of the non-empty ranges are marked in a bit array over the
  // positions of the visit list (this wave's 64 x 64-bit LDS window, rebuilt every 4096
  // positions; the mark of a range that starts at position p > 0 is bit p - 1): the range of
  // position 64 w + l is the number of ranges that start at or before 64 w (one compare against
  // the per-lane offsets + popcount, scalar) plus the marks of positions 64 w + 1 .. 64 w + l
  // (v_mbcnt over the window word) — one LDS read and one ds_bpermute per 64 entries (three for
  // a pair: the two gate penalties), whatever the number of ranges.
  // one load group: NW words located, their loads issued back to back, then tested.  NW is a
  // compile-time count: each group size is straight-line code (branches around loads would
  // make the compiler wait for earlier loads before every later one).
  const bool marks = (u32)lane <= pl.n && pl.offc != 0u;
  auto window = [&](u32 w_first) {     // w_first: a multiple of 64 words
    bits[lane] = 0;
    const u32 wr = ((pl.offc - 1u) >> 6) - w_first;
    if (marks && wr < 64u) atomicOr(reinterpret_cast<unsigned long long *>(bits + wr), 1ull << ((pl.offc - 1u) & 63u));
    __builtin_amdgcn_wave_barrier();
  };
  auto group = [&](auto nw_tag, u32 w0) {
    constexpr int NW = decltype(nw_tag)::value;
    float4 v[NW];
    u32 cellv[NW];    // cell (0..26) of the lane's entry (diagnostic sweep)
    f32x2 pen[NW];    // pair: 0 / +inf per descriptor (the entry's cell passes its gate or not)
    bool valid[NW];   // diagnostic sweep (the exact test reads the cold table, not the sentinel's sides)
#pragma unroll
    for (int u = 0; u < NW; u++) {
      const u32 w_lo = (w0 + u) << 6;
      const u32 pos = w_lo + lane;
      valid[u] = pos < total;
      const u64 bm = bits[(w0 + u) & 63u];      // marks of positions w_lo + 1 .. w_lo + 64
      const u32 before = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pl.offc <= w_lo)) - 1u;
      const u32 j4 = __builtin_amdgcn_mbcnt_hi((u32)(bm >> 32), __builtin_amdgcn_mbcnt_lo((u32)bm, before)) << 2;
      const u32 dsel = (u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)pl.dlc);
      if (DIAG) cellv[u] = (u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)pl.cellc);
      if constexpr (K == 2) {
        pen[u].x = __uint_as_float((u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)__float_as_uint(pl.penc[0])));
        pen[u].y = __uint_as_float((u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)__float_as_uint(pl.penc[1])));
      }
      const u32 e = pos + dsel;       // beyond the list: the sentinel entries
      const float4 *pa = WIDE ? reinterpret_cast<const float4 *>(T.ent + e)
                              : reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(T.ent) + (e << 4));
      v[u] = *pa;             // s0, s1, s2 (f32), id
    }
#ifdef SGTD_EXP_PHASE
    PH_ADD(1, ph_t);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    PH_ADD(2, ph_t);
#endif
    u32 m_start[K];
#pragma unroll
    for (int k = 0; k < K; k++) m_start[k] = matches[k];
    u64 amb_any = 0;     // wave mask: some entry of the group fell between the two f32 thresholds
    // squared f32 distances of word u's entries to the K descriptors; the pair's two sums run
    // in the halves of packed f32 operations and start from the gate penalties
    auto dist2 = [&](int u, float (&d2)[K]) {
      if constexpr (K == 2) {
        const f32x2 qx = {q0f[0], q0f[1]}, qy = {q1f[0], q1f[1]}, qz = {q2f[0], q2f[1]};
        const f32x2 sx = {v[u].x, v[u].x}, sy = {v[u].y, v[u].y}, sz = {v[u].z, v[u].z};
        const f32x2 dx = qx - sx, dy = qy - sy, dz = qz - sz;
        f32x2 acc = __builtin_elementwise_fma(dx, dx, pen[u]);      // pen 0: fl(dx * dx) as in f32_bounds
        acc = __builtin_elementwise_fma(dy, dy, acc);
        acc = __builtin_elementwise_fma(dz, dz, acc);
        d2[0] = acc.x; d2[1] = acc.y;
      } else {
        const float dx = q0f[0] - v[u].x, dy = q1f[0] - v[u].y, dz = q2f[0] - v[u].z;
        d2[0] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      }
    };
    // one (word, descriptor) test; PUSH = false: store the matches, true: replay of the group
    // that only queues the provisional records (rare)
    auto test = [&](auto push_tag, int u, int k, float d2, u32 &count) {
      constexpr bool PUSH = decltype(push_tag)::value;
      const u32 id = __float_as_uint(v[u].w);
      // unsigned (src.frame_id_ - db.frame_id_) > 0  <=>  frame ids differ (:373)
      const bool other = qframe[k] != (id >> id_bits);
      bool hit, amb = false;
      u64 m;
      double dis = 0.0;
      if constexpr (DIAG) {   // the reference's form verbatim on the exact sides, :374-378
        hit = false;
        if (valid[u] && other) {
          const double *sp = T.cold_side + (size_t)id_entry(T.map, id) * 3;
          const double ex = f.q0[k] - sp[0], ey = f.q1[k] - sp[1], ez = f.q2[k] - sp[2];
          dis = sqrt((ex * ex + ey * ey) + ez * ez);   // Eigen norm() association
          hit = dis < thr;
        }
        m = __builtin_amdgcn_ballot_w64(hit);
      } else {
        // (sentinel entries and gated-out cells: d2 = +inf, above every hi2 — f32_bounds keeps it finite)
        const bool near = !(d2 > hi2[k]);     // not certainly outside (NaN stays in)
        hit = near && other;
        amb = hit && !(d2 < lo2[k]);          // not certainly inside either: provisional
        m = __builtin_amdgcn_ballot_w64(near) & __builtin_amdgcn_ballot_w64(other);   // two plain compares: no mask round trip
      }
      const u32 at = count + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
      if constexpr (!PUSH) {
        if (hit && fits) {
          // wave-uniform base (the descriptor's list) + a 32-bit lane offset: no 64-bit VALU address math
          *reinterpret_cast<u32 *>(reinterpret_cast<char *>(list[k]) + (at << 2)) = id;
          if (DIAG) { B.rec_cell[(size_t)slab.next[k] + at] = (unsigned char)cellv[u]; B.rec_dis[(size_t)slab.next[k] + at] = dis; }
        }
        if (!DIAG) amb_any |= m & __builtin_amdgcn_ballot_w64(!(d2 < lo2[k]));
      } else {
        if (amb && fits) {
          const u32 qa = atomicAdd(B.amb_count, 1u);
          if (qa < B.amb_cap) B.amb_queue[qa] = make_uint2(slab.next[k] + at, f.slot[k]);
          else B.overflow[0] = 1;    // re-run with a larger queue (grows with the record buffer)
        }
      }
      count += (u32)__builtin_popcountll(m);
    };
#pragma unroll
    for (int u = 0; u < NW; u++) {
      float d2[K];
      if (!DIAG) dist2(u, d2);
#pragma unroll
      for (int k = 0; k < K; k++) test(std::false_type{}, u, k, DIAG ? 0.0f : d2[k], matches[k]);
    }
    if (!DIAG && amb_any) {   // rare: about one in 10^4 matches
#pragma unroll
      for (int u = 0; u < NW; u++) {
        float d2[K];
        dist2(u, d2);
#pragma unroll
        for (int k = 0; k < K; k++) test(std::true_type{}, u, k, d2[k], m_start[k]);
      }
    }
    PH_ADD(3, ph_t);
  };
  // full groups of SGTD_PROBE_UNROLL words, window after window, then what is left in groups
  // of 2 and 1 (inside the last window)
  if (n_words) {
    u32 w0 = 0;
    bool touched = false;
    while (true) {
      window(w0);
      const u32 w_stop = min(n_words, w0 + 64u);
      for (; w0 + SGTD_PROBE_UNROLL <= w_stop; w0 += SGTD_PROBE_UNROLL) {
        group(std::integral_constant<int, SGTD_PROBE_UNROLL>{}, w0);
        if (!touched) { pending.touch(); touched = true; }
      }
      if (w_stop == n_words) break;
    }
    const u32 left = n_words - w0;   // wave-uniform
    if (left & 2u) { group(std::integral_constant<int, 2>{}, w0); w0 += 2; }
    if (left & 1u) { group(std::integral_constant<int, 1>{}, w0); w0 += 1; }
    if (!touched) pending.touch();   // every path through the sweep leaves them complete
  } else {
    pending.touch();
  }
  static_assert(SGTD_PROBE_UNROLL == 4, "remainder groups cover 2 and 1 words; a window is a whole number of groups");
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (!fits && lane == 0) atomicAdd(B.rec_need, (unsigned long long)matches[k]);
    result[k].ptr = slab.next[k]; result[k].visit = pl.ref_visits[k]; result[k].match = fits ? matches[k] : 0;
    if (fits) slab.next[k] += matches[k];
  }
  __builtin_amdgcn_wave_barrier();
  PH_ADD(4, ph_t);
}

#ifndef SGTD_SWEEP_OCC
#define SGTD_SWEEP_OCC
#endif
#define SGTD_TICKET_MAX 16   // descriptors per ticket: 4 lanes each in one 64-lane load
#define SGTD_NO_CHUNK 0xFFFFFFFFu

// The per-XCD ticket queues of the sweep: chunk ids [c_lo, c_hi) of queue x belong
// to XCD x; a wave drains its own XCD's queue first, then helps the others (their ranges
// lose locality but keep the chip busy).  Heads are 4 KB apart (own L2 channel each).
struct TicketQueue {
  u32 *heads;
  u32 n_chunks, xcc;
  u32 t, c_lo, c_hi;     // current queue = (xcc + t) & 7
  __device__ __forceinline__ void select(u32 t_) {
    t = t_;
    const u32 x = (xcc + t) & 7u;
    c_lo = (u32)(((u64)n_chunks * x) >> 3);
    c_hi = (u32)(((u64)n_chunks * (x + 1)) >> 3);
  }
  // one ticket of the current queue, not waited for (lane 0 holds it)
  __device__ __forceinline__ u32 issue() const {
    u32 tk = 0;
    if (t < 8 && lane_id() == 0) tk = atomicAdd(&heads[((xcc + t) & 7u) * 1024u], 1u);
    return tk;
  }
  // ticket -> chunk id; moves on to the next queues (blocking) when this one is drained
  __device__ __forceinline__ u32 resolve(u32 tk) {
    while (t < 8) {
      const u32 c = c_lo + (u32)__builtin_amdgcn_readfirstlane((int)tk);
      if (c < c_hi) return c;
      select(t + 1);
      tk = issue();
    }
    return SGTD_NO_CHUNK;
  }
};

template <bool DIAG, bool WIDE>
__global__ __launch_bounds__(SGTD_PROBE_THREADS) SGTD_SWEEP_OCC void probe_sorted_kernel(
    TableView T, ProbeBuffers B, const unsigned char *rows, const QueryRec *sorted, double rough,
    const u32 *n_valid_p, u32 *xcd_heads /*[8 * 1024]*/, u32 chunk /* 1..SGTD_TICKET_MAX */) {
  __shared__ u64 s_bits[SGTD_PROBE_THREADS / SGTD_WAVE][64];   // per wave: range starts of the current 4096 positions
  const int lane = lane_id();
  const u32 n_valid = *n_valid_p;
  // Every WAVE dequeues `chunk` consecutive sorted positions at a time (no workgroup
  // barrier).  Small chunks keep the descriptors in flight on one XCD — and with them the
  // buckets it is reading — within that XCD's 4 MB L2; the host sizes a ticket to about 2k
  // entry visits.  Software pipeline per wave: the ticket after next is in flight, the next
  // ticket's descriptor records are in flight, the next descriptor's GroupRow is in flight
  // while the current descriptor is swept.
  TicketQueue tq;
  tq.heads = xcd_heads;
  tq.n_chunks = (n_valid + chunk - 1) / chunk;
  u32 xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  tq.xcc = xcc & 7u;
  tq.select(0);
  WaveSlab slab{};
#ifdef SGTD_EXP_PHASE
  for (int i = 0; i < 8; i++) slab.ph[i] = 0;
  const u64 ph_start = PH_T();
#endif
#ifdef SGTD_EXP_TRACE
  const u64 tr_t0 = wall_clock64();
  u64 tr_own = 0; u32 tr_n_own = 0, tr_n_st = 0;
#endif
  // lane j loads quarter j & 3 of descriptor j >> 2 of the chunk
  auto load_chunk = [&](u32 c) {
    const u32 p = c * chunk + ((u32)lane >> 2);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (c != SGTD_NO_CHUNK && ((u32)lane >> 2) < chunk && p < n_valid)
      r = reinterpret_cast<const uint4 *>(sorted + p)[lane & 3];
    return r;
  };
  // lane l < 54 loads 16-B quarter l of the group's 27 directory rows
  auto load_row = [&](u32 g) {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (lane < SGTD_NRANGE) r = reinterpret_cast<const uint4 *>(rows + (size_t)g * SGTD_GROUP_ROW_BYTES)[lane];
    return r;
  };

  u32 cur_c = tq.resolve(tq.issue());
  uint4 rec = load_chunk(cur_c);
  asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(rec.z), "+v"(rec.w));   // same for the ticket loop
  u32 tk_next = tq.issue();
  while (cur_c != SGTD_NO_CHUNK) {
    const u32 nxt_c = tq.resolve(tk_next);        // requested one whole chunk ago
    tk_next = tq.issue();                         // in flight during this chunk
    uint4 rec_next = load_chunk(nxt_c);           // in flight during this chunk
    const u32 p_first = cur_c * chunk;
    const u32 n = min(chunk, n_valid - p_first);
#ifdef SGTD_EXP_TRACE
    if (tq.t == 0) tr_n_own += n; else { if (!tr_own) tr_own = wall_clock64(); tr_n_st += n; }
#endif
    u32 g_cur = (u32)__builtin_amdgcn_readlane((int)rec.z, 2);
    uint4 row_next = load_row(g_cur);
    // waited for here, once per ticket: the loop below then carries no pending load into its
    // header on either edge (the next rows are touched inside the sweep)
    asm volatile("" : "+v"(row_next.x), "+v"(row_next.y), "+v"(row_next.z), "+v"(row_next.w));
    u32 r_ptr = 0, r_visit = 0, r_match = 0;   // lane i: results of descriptor i of the chunk
    // descriptor i of the chunk -> its fields (records are 4 lanes each)
    auto unpack = [&](auto &f, int k, u32 i) {
      const int l0 = (int)(4 * i);
      f.q0[k] = __hiloint2double(__builtin_amdgcn_readlane((int)rec.y, l0), __builtin_amdgcn_readlane((int)rec.x, l0));
      f.q1[k] = __hiloint2double(__builtin_amdgcn_readlane((int)rec.w, l0), __builtin_amdgcn_readlane((int)rec.z, l0));
      f.q2[k] = __hiloint2double(__builtin_amdgcn_readlane((int)rec.y, l0 + 1), __builtin_amdgcn_readlane((int)rec.x, l0 + 1));
      f.thr2[k] = __hiloint2double(__builtin_amdgcn_readlane((int)rec.w, l0 + 1), __builtin_amdgcn_readlane((int)rec.z, l0 + 1));
      f.qframe[k] = (u32)__builtin_amdgcn_readlane((int)rec.x, l0 + 2);
      f.gate[k] = (u32)__builtin_amdgcn_readlane((int)rec.y, l0 + 2);
      f.slot[k] = (u32)__builtin_amdgcn_readlane((int)rec.w, l0 + 2);
      f.lo2[k] = __uint_as_float((u32)__builtin_amdgcn_readlane((int)rec.x, l0 + 3));
      f.hi2[k] = __uint_as_float((u32)__builtin_amdgcn_readlane((int)rec.y, l0 + 3));
      f.t_up[k] = __uint_as_float((u32)__builtin_amdgcn_readlane((int)rec.z, l0 + 3));
    };
    for (u32 i = 0; i < n;) {
      // consecutive descriptors of one home cell are swept together, 4 or 2 at a time (one plan,
      // one locate and one load per 64 entries for all of them); the diagnostic build takes them
      // one by one
      u32 run = 1;
      if (!DIAG)
        while (run < (u32)SGTD_PAIR && i + run < n &&
               (u32)__builtin_amdgcn_readlane((int)rec.z, (int)(4 * (i + run) + 2)) == g_cur) run++;
      const u32 step = run >= 2 ? 2u : 1u;
      const uint4 row = row_next;
      if (i + step < n) {   // the GroupRow after this step (often the same one)
        const u32 g1 = (u32)__builtin_amdgcn_readlane((int)rec.z, (int)(4 * (i + step) + 2));
        if (g1 != g_cur) row_next = load_row(g1);
        g_cur = g1;
      }
      PendingLoads pend;
      pend.row = &row_next;
      pend.rec = &rec_next;
      // a home cell none of whose 27 buckets exists in this table segment (common for the small
      // tail segment of an appended map): nothing to plan or sweep
      if (!__builtin_amdgcn_ballot_w64((row.x | row.y | row.z | row.w) != 0u)) {
        pend.touch();
        i += step;
        continue;
      }
      auto pass = [&](auto k_tag) {
        constexpr int KK = decltype(k_tag)::value;
        DescSet<KK> f;
        f.row = row;
#pragma unroll
        for (int k = 0; k < KK; k++) unpack(f, k, i + (u32)k);
        DescResult res[KK];
        sweep_descriptors<DIAG, WIDE, KK>(T, B, rough, f, plan_from_group_row<KK>(f, T.n_entries, T.coarse_at), s_bits[threadIdx.x >> 6], slab, res, pend);
#pragma unroll
        for (int k = 0; k < KK; k++)
          if ((u32)lane == i + (u32)k) { r_ptr = res[k].ptr; r_visit = res[k].visit; r_match = res[k].match; }
      };
      if constexpr (!DIAG && SGTD_PAIR >= 2) { if (step == 2) pass(std::integral_constant<int, 2>{}); }
      if (step == 1) pass(std::integral_constant<int, 1>{});
      i += step;
    }
    {   // the chunk's results: lane i < n stores for its descriptor (slot d in quarter 2 of record i)
      const u32 d_mine = (u32)__shfl((int)rec.w, (4 * lane + 2) & 63);
      if ((u32)lane < n) {
        B.list_ptr[d_mine] = r_ptr;
        B.n_visit[d_mine] = r_visit;
        B.n_match[d_mine] = r_match;
      }
    }
    cur_c = nxt_c;
    rec = rec_next;
  }
  if (lane == 0 && slab.swept) atomicAdd(B.swept, (unsigned long long)slab.swept);
#ifdef SGTD_EXP_PHASE
  if (lane == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(&g_phase[i], slab.ph[i]);
    atomicAdd(&g_phase[7], PH_T() - ph_start);
    atomicAdd(&g_phase[6], 1ull);
  }
#endif
#ifdef SGTD_EXP_TRACE
  if (lane == 0) {
    u64 *tr = reinterpret_cast<u64 *>(xcd_heads + 8 * 1024) + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    tr[0] = tr_t0; tr[1] = tr_own; tr[2] = wall_clock64();
    tr[3] = ((u64)tq.xcc << 56) | ((u64)tr_n_own << 28) | (u64)tr_n_st;
  }
#endif
}

// The provisional records of the sweep, decided exactly (STDesc.cpp:374-378 in the squared,
// comparison-exact form): a record whose entry does not match after all gets the frame
// of no frame (SGTD_DEAD_ID: no vote, no candidate) and leaves the query's match count.
__global__ void resolve_undecided_kernel(TableView T, QueryView Q, ProbeBuffers B, u32 *q_M) {
  const u32 n = min(*B.amb_count, B.amb_cap);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint2 it = B.amb_queue[i];
    const QueryRec &r = Q.qrec[it.y];
    const double *sp = T.cold_side + (size_t)id_entry(T.map, B.rec[it.x]) * 3;
    const double dx = r.q0 - sp[0], dy = r.q1 - sp[1], dz = r.q2 - sp[2];
    const double d2 = (dx * dx + dy * dy) + dz * dz;   // Eigen norm() association
    if (!(d2 < r.thr2)) {
      B.rec[it.x] = SGTD_DEAD_ID;
      atomicSub(&q_M[(u32)((long long)it.y / Q.stride)], 1u);
    }
  }
}

// votes (:404-420) from the match lists: one wavefront per 128-descriptor block,
// the 4 blocks of a workgroup belong to one query and share an LDS histogram
template <bool LDS_VOTES>
__global__ __launch_bounds__(256) void votes_kernel(QueryView Q, ProbeBuffers B, u32 frame_span, u32 frame_lo,
                                                    int blocks_per_query, u32 *q_M, unsigned long long *q_P);

// top candidate_num frames of one query (:423-433): arg-max rounds over
// (votes, lowest frame id), requires votes >= 5; marks slot_of[frame] = slot.
// Fast path: a vote-value histogram gives the threshold below which no frame can
// make the list; the <= 1024 frames at or above it are pooled and one wavefront
// runs the reference's rounds on the pool held in registers.  If more frames tie
// at the threshold than the pool holds, the rounds run over all frames.
#define SGTD_TOPK_BINS 8192
#define SGTD_TOPK_POOL 1024
__global__ __launch_bounds__(256) void topk_kernel(const u32 *votes_all, u32 frame_span, u32 frame_lo,
                                                   int cand_num, int *n_cand, int *cand_frame,
                                                   int *cand_votes, unsigned char *slot_of_all) {
  __shared__ u64 red[256 / SGTD_WAVE];
  __shared__ int n_picked;
  __shared__ u32 s_hist[SGTD_TOPK_BINS];
  __shared__ u64 s_pool[SGTD_TOPK_POOL];
  __shared__ u32 s_thr, s_n_ge, s_npool;
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = lane_id();
  const u32 *votes = votes_all + (size_t)q * frame_span;
  unsigned char *slot_of = slot_of_all + (size_t)q * frame_span;
  for (int b = tid; b < SGTD_TOPK_BINS; b += 256) s_hist[b] = 0;
  if (tid == 0) { n_picked = 0; s_npool = 0; }
  __syncthreads();
  for (u32 f = tid; f < frame_span; f += 256) {
    const u32 v = votes[f];
    if (v >= 5) atomicAdd(&s_hist[min(v, (u32)SGTD_TOPK_BINS - 1u)], 1u);
  }
  __syncthreads();
  if (tid < SGTD_WAVE) {
    // largest t with count(votes >= t) >= cand_num (t = 5 if fewer frames qualify at all)
    constexpr int PER = SGTD_TOPK_BINS / SGTD_WAVE;
    u32 mine = 0;
    for (int b = 0; b < PER; b++) mine += s_hist[lane * PER + b];
    u32 suffix = mine;   // inclusive suffix sum over lanes >= lane
#pragma unroll
    for (int d = 1; d < SGTD_WAVE; d <<= 1) {
      const u32 o = __shfl_down(suffix, d);
      if (lane + d < SGTD_WAVE) suffix += o;
    }
    const u64 okmask = __ballot(suffix >= (u32)cand_num);
    u32 thr = 5, n_ge = __shfl(suffix, 0);
    if (okmask) {
      const int L = 63 - __builtin_clzll(okmask);
      if (lane == L) {
        u32 acc = suffix - mine;
        int b = L * PER + PER - 1;
        for (; b >= L * PER; b--) { acc += s_hist[b]; if (acc >= (u32)cand_num) break; }
        s_thr = (u32)max(b, 5); s_n_ge = acc;
      }
    } else if (lane == 0) { s_thr = thr; s_n_ge = n_ge; }
  }
  __syncthreads();
  const u32 thr = s_thr;
  if (s_n_ge <= (u32)SGTD_TOPK_POOL) {
    for (u32 f = tid; f < frame_span; f += 256) {
      const u32 v = votes[f];
      if (v >= thr) s_pool[atomicAdd(&s_npool, 1u)] = ((u64)v << 32) | (u64)(0xFFFFFFFFu - f);
    }
    __syncthreads();
    if (tid < SGTD_WAVE) {
      constexpr int PER = SGTD_TOPK_POOL / SGTD_WAVE;
      const u32 np = s_npool;
      u64 a[PER];
#pragma unroll
      for (int k = 0; k < PER; k++) a[k] = ((u32)(k * SGTD_WAVE + lane) < np) ? s_pool[k * SGTD_WAVE + lane] : 0ull;
      int picked = 0;
      for (int round = 0; round < cand_num; round++) {
        u64 best = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) best = a[k] > best ? a[k] : best;
#pragma unroll
        for (int dlt = SGTD_WAVE / 2; dlt > 0; dlt >>= 1) {
          const u64 o = __shfl_xor(best, dlt);
          best = o > best ? o : best;
        }
        if ((u32)(best >> 32) < 5u) break;   // max_vote > 1 && max_vote >= 5 (:427,433)
#pragma unroll
        for (int k = 0; k < PER; k++) a[k] = (a[k] == best) ? 0ull : a[k];   // match_array[...] = 0 (:435)
        if (lane == 0) {
          const u32 f = 0xFFFFFFFFu - (u32)(best & 0xFFFFFFFFu);
          slot_of[f] = (unsigned char)picked;
          cand_frame[q * cand_num + picked] = (int)(f + frame_lo);
          cand_votes[q * cand_num + picked] = (int)(u32)(best >> 32);
        }
        picked++;
      }
      if (lane == 0) n_cand[q] = picked;
    }
    return;
  }
  // ---- general path: more ties at the threshold than the pool holds
  for (int round = 0; round < cand_num; round++) {
    // key = votes << 32 | ~local frame : max key = most votes, then lowest frame
    u64 best = 0;
    for (u32 f = threadIdx.x; f < frame_span; f += 256) {
      if (slot_of[f] != 0xFF) continue;   // already taken (its match_array entry was zeroed, :435)
      u64 key = ((u64)votes[f] << 32) | (u64)(0xFFFFFFFFu - f);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int dlt = SGTD_WAVE / 2; dlt > 0; dlt >>= 1) {
      u64 o = __shfl_xor(best, dlt);
      best = o > best ? o : best;
    }
    if (lane_id() == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
      u64 b = red[0];
      for (int w = 1; w < 256 / SGTD_WAVE; w++) b = red[w] > b ? red[w] : b;
      const u32 v = (u32)(b >> 32);
      if (v >= 5) {   // max_vote > 1 && max_vote >= 5 (:427,433)
        const u32 f = 0xFFFFFFFFu - (u32)(b & 0xFFFFFFFFu);
        slot_of[f] = (unsigned char)n_picked;
        cand_frame[q * cand_num + n_picked] = (int)(f + frame_lo);
        cand_votes[q * cand_num + n_picked] = (int)v;
        n_picked++;
      } else {
        n_picked |= 0x40000000;  // stop marker
      }
      __threadfence_block();
    }
    __syncthreads();
    if (n_picked & 0x40000000) break;
  }
  if (threadIdx.x == 0) n_cand[q] = n_picked & 0x3FFFFFFF;
}

// ---------------------------------------------------------------------------
// assemble: one wavefront per block of SGTD_PROBE_CHUNK query descriptors.
// Workgroup b -> (query, group of 4 blocks) with b % 8 == query % 8, so that all
// workgroups of a query run on one XCD (workgroups are dealt round-robin over
// the 8 XCDs) and the partial output lines of neighbouring blocks merge in
// that XCD's L2 — a speed choice only, results do not depend on placement.
// ---------------------------------------------------------------------------
struct BlockId {
  int q;          // query
  int blk;        // 128-descriptor block inside the query
  bool valid;
};

__device__ __forceinline__ BlockId assemble_block(int n_queries, int blocks_per_query) {
  constexpr int NW = 256 / SGTD_WAVE;
  const int groups = (blocks_per_query + NW - 1) / NW;       // workgroups per query
  const int b = blockIdx.x, x = b & 7, r = b >> 3;
  BlockId id;
  id.q = (r / groups) * 8 + x;
  id.blk = (r % groups) * NW + (int)(threadIdx.x >> 6);
  id.valid = id.q < n_queries && id.blk < blocks_per_query;
  return id;
}

// the 32-descriptor sub-block [d0, d0+32) of query q: prefix of n_match and list
// pointers into LDS; returns the number of records
__device__ __forceinline__ u32 sub_open(const QueryView &Q, const ProbeBuffers &B, int sg, int q, u32 d0, u32 cnt,
                                        u32 *s_pre /*[32]*/, u32 *s_ptr /*[32]*/, u32 &visits) {
  const int lane = lane_id();
  u32 n = 0, p = 0, v = 0;
  if (lane < SGTD_SUB_DESCS && d0 + lane < cnt) {
    const long long d = (long long)sg * B.seg_stride + (long long)q * Q.stride + d0 + lane;
    n = B.n_match[d]; p = B.list_ptr[d]; v = B.n_visit[d];
  }
  const u32 inc = wave_incl_scan(n);
  const u32 R = __shfl(inc, SGTD_WAVE - 1);
  visits += wave_sum(v);
  __builtin_amdgcn_wave_barrier();
  if (lane < 32) { s_pre[lane] = inc - n; s_ptr[lane] = p; }
  __builtin_amdgcn_wave_barrier();
  return R;
}

// record r of the sub-block -> (descriptor index inside it, record address)
__device__ __forceinline__ void sub_locate(const u32 *s_pre, const u32 *s_ptr, u32 r, u32 &dd, u32 &addr) {
  // last dd with pre[dd] <= r (descriptors without matches share offsets and are skipped)
  u32 c = 0;
  if (s_pre[16] <= r) c = 16;
  if (s_pre[c + 8] <= r) c += 8;
  if (s_pre[c + 4] <= r) c += 4;
  if (s_pre[c + 2] <= r) c += 2;
  if (s_pre[c + 1] <= r) c += 1;
  dd = c;
  addr = s_ptr[c] + (r - s_pre[c]);
}

// The same sub-block as a stream of QUADS — four consecutive records of one list, the last quad
// of a list possibly short: one list search serves four records (the record buffer has room
// for the reads past a list's end).  s_pre: exclusive quad offsets, s_ptr: list starts, s_cnt:
// list lengths; returns the number of quads, `records` the number of records.
__device__ __forceinline__ u32 sub_open_quads(const QueryView &Q, const ProbeBuffers &B, int sg, int q, u32 d0, u32 cnt,
                                              u32 *s_pre /*[32]*/, u32 *s_ptr /*[32]*/, u32 *s_cnt /*[32]*/,
                                              u32 &visits, u32 &records) {
  const int lane = lane_id();
  u32 n = 0, p = 0, v = 0;
  if (lane < SGTD_SUB_DESCS && d0 + lane < cnt) {
    const long long d = (long long)sg * B.seg_stride + (long long)q * Q.stride + d0 + lane;
    n = B.n_match[d]; p = B.list_ptr[d]; v = B.n_visit[d];
  }
  const u32 nq = (n + 3u) >> 2;
  const u32 inc = wave_incl_scan(nq);
  const u32 RQ = __shfl(inc, SGTD_WAVE - 1);
  visits += wave_sum(v);
  records = wave_sum(n);
  __builtin_amdgcn_wave_barrier();
  if (lane < 32) { s_pre[lane] = inc - nq; s_ptr[lane] = p; s_cnt[lane] = n; }
  __builtin_amdgcn_wave_barrier();
  return RQ;
}

// quad r of the sub-block -> (descriptor index inside it, address of its first record, records in it)
__device__ __forceinline__ void sub_locate_quad(const u32 *s_pre, const u32 *s_ptr, const u32 *s_cnt, u32 r, u32 &dd,
                                                u32 &addr, u32 &k) {
  u32 c = 0;
  if (s_pre[16] <= r) c = 16;
  if (s_pre[c + 8] <= r) c += 8;
  if (s_pre[c + 4] <= r) c += 4;
  if (s_pre[c + 2] <= r) c += 2;
  if (s_pre[c + 1] <= r) c += 1;
  dd = c;
  const u32 first = (r - s_pre[c]) << 2;      // record index of the quad inside its list
  addr = s_ptr[c] + first;
  k = min(4u, s_cnt[c] - first);
}

// frame -> candidate slot of one query as an LDS open-addressing table (cand_num <= 64
// frames in 256 slots): the assemble kernels look every match record up here instead of in
// the per-frame slot_of array in global memory — no dependent global load per record, and
// the LDS footprint does not grow with the number of map frames.  Entry = frame << 8 | slot.
#define SGTD_CAND_HASH 256
#define SGTD_CAND_EMPTY 0xFFFFFFFFFFFFFFFFull
// all 256 threads of the workgroup (blockDim.x == SGTD_CAND_HASH)
__device__ __forceinline__ void cand_hash_build(u64 *s_tab, const int *n_cand, const int *cand_frame, int q,
                                                int n_queries, int cand_num) {
  s_tab[threadIdx.x] = SGTD_CAND_EMPTY;
  __syncthreads();
  const int s = (int)threadIdx.x;
  if (q < n_queries && s < cand_num && s < n_cand[q]) {
    const u32 f = (u32)cand_frame[(size_t)q * cand_num + s];
    u32 h = (f * 0x9E3779B1u) >> 24;
    while (atomicCAS(&s_tab[h], SGTD_CAND_EMPTY, ((u64)f << 8) | (u64)s) != SGTD_CAND_EMPTY) h = (h + 1) & (SGTD_CAND_HASH - 1);
  }
  __syncthreads();
}

__device__ __forceinline__ u32 cand_slot(const u64 *s_tab, u32 frame) {
  u32 h = (frame * 0x9E3779B1u) >> 24;
  while (true) {
    const u64 e = s_tab[h];
    if (e == SGTD_CAND_EMPTY) return 0xFFu;
    if ((u32)(e >> 8) == frame) return (u32)e & 0xFFu;
    h = (h + 1) & (SGTD_CAND_HASH - 1);
  }
}

// the match lists of one 128-descriptor block counted into a vote histogram (LDS or global)
// (bins = local frames [frame_lo, frame_lo + limit): records of other frames — another tile's, or dead ones — are skipped)
template <bool LDS_VOTES>
__device__ __forceinline__ void votes_of_block(const QueryView &Q, const ProbeBuffers &B, int q, u32 d_first, u32 cnt,
                                               u32 frame_lo, u32 limit, u32 *s_hist, u32 *votes, u32 *s_pre, u32 *s_ptr,
                                               u32 *s_cnt, u32 &visits, u32 &total) {
  const int lane = lane_id();
  for (int sg = 0; sg < B.n_seg; sg++)
  for (u32 d0 = d_first; d0 < min(d_first + SGTD_PROBE_CHUNK, cnt); d0 += SGTD_SUB_DESCS) {
    u32 records;
    const u32 RQ = sub_open_quads(Q, B, sg, q, d0, cnt, s_pre, s_ptr, s_cnt, visits, records);
    total += records;
    // the quads of the next two words are loaded while the current two are counted
    uint4 nrec[SGTD_VOTE_WORDS];
    u32 nk[SGTD_VOTE_WORDS];
    auto load2 = [&](u32 r0) {
#pragma unroll
      for (int u = 0; u < SGTD_VOTE_WORDS; u++) {
        const u32 r = r0 + u * SGTD_WAVE + lane;
        u32 dd, addr, k;
        sub_locate_quad(s_pre, s_ptr, s_cnt, r < RQ ? r : 0u, dd, addr, k);
        nk[u] = r < RQ ? k : 0u;
        const u32 *src = B.rec + addr;      // 4-byte aligned
        nrec[u] = make_uint4(src[0], src[1], src[2], src[3]);
      }
    };
    if (RQ) load2(0);
    for (u32 r0 = 0; r0 < RQ; r0 += SGTD_VOTE_WORDS * SGTD_WAVE) {
      uint4 rc[SGTD_VOTE_WORDS];
      u32 kk[SGTD_VOTE_WORDS];
#pragma unroll
      for (int u = 0; u < SGTD_VOTE_WORDS; u++) { rc[u] = nrec[u]; kk[u] = nk[u]; }
      if (r0 + SGTD_VOTE_WORDS * SGTD_WAVE < RQ) load2(r0 + SGTD_VOTE_WORDS * SGTD_WAVE);
#pragma unroll
      for (int u = 0; u < SGTD_VOTE_WORDS; u++) {
        const u32 w4[4] = {rc[u].x, rc[u].y, rc[u].z, rc[u].w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const u32 bin = (w4[i] >> B.id_bits) - frame_lo;     // local frame (a dead record's is beyond every span)
          if ((u32)i < kk[u] && bin < limit) {
            if (LDS_VOTES) atomicAdd(&s_hist[bin], 1u);
            else atomicAdd(&votes[bin], 1u);
          }
        }
      }
    }
  }
}

template <bool LDS_VOTES>
__global__ __launch_bounds__(256) void votes_kernel(QueryView Q, ProbeBuffers B, u32 frame_span, u32 frame_lo,
                                                    int blocks_per_query, u32 *q_M, unsigned long long *q_P) {
  constexpr int NW = 256 / SGTD_WAVE;
  extern __shared__ u32 s_hist[];   // [frame_span] when LDS_VOTES
  __shared__ u32 s_pre[NW][32];
  __shared__ u32 s_ptr[NW][32];
  __shared__ u32 s_cnt[NW][32];
  if (B.overflow[0]) return;
  const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
  const BlockId id = assemble_block(Q.n_queries, blocks_per_query);
  if (id.q >= Q.n_queries) return;   // workgroup-uniform: all 4 waves share the query
  const int q = id.q;
  if (LDS_VOTES) {
    for (u32 f = tid; f < frame_span; f += 256) s_hist[f] = 0;
    __syncthreads();
  }
  u32 *votes = B.votes + (size_t)q * frame_span;
  const u32 cnt = Q.count[q];
  const u32 d_first = (u32)id.blk * SGTD_PROBE_CHUNK;
  if (id.valid && d_first < cnt) {
    u32 visits = 0, total = 0;
    votes_of_block<LDS_VOTES>(Q, B, q, d_first, cnt, 0u, frame_span, s_hist, votes, s_pre[wid], s_ptr[wid], s_cnt[wid], visits, total);
    if (lane == 0) {
      atomicAdd(&q_M[q], total);
      atomicAdd(&q_P[q], (unsigned long long)visits);
    }
  }
  if (LDS_VOTES) {
    __syncthreads();
    for (u32 f = tid; f < frame_span; f += 256) {
      const u32 v = s_hist[f];
      if (v) atomicAdd(&votes[f], v);
    }
  }
}

// votes of a whole query by ONE workgroup of 16 wavefronts (batches with enough queries to fill
// the chip that way): the LDS histogram is the query's final vote array — plain coalesced stores,
// no global atomics, no pre-zeroed vote buffer.  A frame span beyond LDS is cut into tiles of
// tile_span bins, one workgroup per (query, tile) = blockIdx.(x, y): every tile's workgroup walks
// all of the query's records and counts the ones of its frames.
#define SGTD_VOTES_Q_THREADS 1024
__global__ __launch_bounds__(SGTD_VOTES_Q_THREADS) void votes_query_kernel(QueryView Q, ProbeBuffers B, u32 frame_span,
                                                                            u32 frame_lo, u32 tile_span, int blocks_per_query,
                                                                            u32 *q_M, unsigned long long *q_P) {
  constexpr int NW = SGTD_VOTES_Q_THREADS / SGTD_WAVE;
  extern __shared__ u32 s_hist[];   // [tile_span]
  __shared__ u32 s_pre[NW][32];
  __shared__ u32 s_ptr[NW][32];
  __shared__ u32 s_cnt[NW][32];
  __shared__ u32 s_M;
  __shared__ unsigned long long s_P;
  const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
  const int q = blockIdx.x;
  const u32 tile_lo = blockIdx.y * tile_span;
  const u32 n_bins = min(tile_span, frame_span - tile_lo);
  const bool dead = B.overflow[0] != 0;     // the batch is re-run: leave zeros
  for (u32 f = tid; f < n_bins; f += SGTD_VOTES_Q_THREADS) s_hist[f] = 0;
  if (tid == 0) { s_M = 0; s_P = 0; }
  __syncthreads();
  const u32 cnt = Q.count[q];
  if (!dead) {
    u32 visits = 0, total = 0;
    for (int blk = wid; blk < blocks_per_query; blk += NW) {
      const u32 d_first = (u32)blk * SGTD_PROBE_CHUNK;
      if (d_first >= cnt) break;
      votes_of_block<true>(Q, B, q, d_first, cnt, tile_lo, n_bins, s_hist, nullptr, s_pre[wid], s_ptr[wid], s_cnt[wid], visits, total);
    }
    if (lane == 0 && blockIdx.y == 0) {
      atomicAdd(&s_M, total);
      atomicAdd(&s_P, (unsigned long long)visits);
    }
  }
  __syncthreads();
  u32 *votes = B.votes + (size_t)q * frame_span + tile_lo;
  for (u32 f = tid; f < n_bins; f += SGTD_VOTES_Q_THREADS) votes[f] = s_hist[f];
  if (tid == 0 && blockIdx.y == 0) {      // resolve_undecided_kernel has already subtracted the records it killed
    atomicAdd(&q_M[q], s_M);
    atomicAdd(&q_P[q], s_P);
  }
}

// pass 1: blk_count[(q*blocks+blk)*64 + s] = matches of the block in slot s;
// also the per-query sums of visited entries / matches for the statistics
// compact list of one query's candidate matches, block after block in list order: the pairs
// (q_idx << 32 | g) of the records whose frame made the candidate list, and their slots.
// Written by block_count_kernel, consumed by block_write_kernel (no second record walk).
struct CompactLists {
  u64 *pair;             // [cap] slot << 58 | q_idx << 32 | entry id
  u32 *blk_start;        // [nq * blocks_per_query] first entry of the block's list
  u32 *blk_n;            // [nq * blocks_per_query] entries of the block's list
  u32 *cursor;           // global allocation cursor
  u32 cap;
};

// SLOT_TABLE: the frame -> candidate slot map of the query is the byte array topk_kernel wrote
// (slot_of, 0xFF = not a candidate), staged in LDS and indexed directly — one ds_read_u8 per
// record; for maps whose frame span does not fit LDS the 256-entry hash of the candidates
template <bool SLOT_TABLE>
__global__ __launch_bounds__(256) void block_count_kernel(QueryView Q, ProbeBuffers B, const int *n_cand,
                                                          const int *cand_frame, int cand_num,
                                                          int blocks_per_query, u32 *blk_count, CompactLists L,
                                                          u32 *q_M, unsigned long long *q_P,
                                                          const unsigned char *slot_of_all, u32 frame_span, u32 frame_lo) {
  constexpr int NW = 256 / SGTD_WAVE;
  extern __shared__ unsigned char s_slot8[];   // [frame_span rounded up to 16] when SLOT_TABLE
  __shared__ u32 s_pre[NW][32];
  __shared__ u32 s_ptr[NW][32];
  __shared__ u32 s_hist[NW][64];
  __shared__ u64 s_cand[SGTD_CAND_HASH];
  if (B.overflow[0]) return;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const BlockId id = assemble_block(Q.n_queries, blocks_per_query);
  const int q = id.q;
  if (SLOT_TABLE) {
    if (q < Q.n_queries) {   // workgroup-uniform
      const unsigned char *src = slot_of_all + (size_t)q * frame_span;
      for (u32 f = threadIdx.x; f < frame_span; f += 256) s_slot8[f] = src[f];
    }
    __syncthreads();
  } else {
    cand_hash_build(s_cand, n_cand, cand_frame, q, Q.n_queries, cand_num);
  }
  if (!id.valid) return;
  const u32 cnt = Q.count[q];
  const u32 d_first = (u32)id.blk * SGTD_PROBE_CHUNK;
  const size_t bslot = (size_t)q * blocks_per_query + id.blk;
  u32 *out = blk_count + bslot * 64;
  if (d_first >= cnt) { out[lane] = 0; if (lane == 0) { L.blk_start[bslot] = 0; L.blk_n[bslot] = 0; } return; }
  const u32 d_last = min(d_first + SGTD_PROBE_CHUNK, cnt);
  // room for the block's compact list: at most every record of the block
  u32 nm = 0;
  for (int sg = 0; sg < B.n_seg; sg++)
    for (u32 dd = d_first + lane; dd < d_last; dd += SGTD_WAVE) nm += B.n_match[(long long)sg * B.seg_stride + (long long)q * Q.stride + dd];
  const u32 r_blk = wave_sum(nm);
  u32 start = 0;
  if (lane == 0) start = atomicAdd(L.cursor, r_blk);
  start = (u32)__builtin_amdgcn_readfirstlane((int)start);
  const bool fits = (unsigned long long)start + r_blk <= (unsigned long long)L.cap;
  if (!fits && lane == 0) B.overflow[0] = 1;     // sized like the record buffer: grown and re-run with it
  s_hist[wid][lane] = 0;
  u32 visits = 0, total = 0, running = 0;
  // segment by segment: a candidate frame lives in one segment, so its matches still arrive in
  // (i, cell, j) order
  for (int sg = 0; sg < B.n_seg; sg++)
  for (u32 d0 = d_first; d0 < d_last; d0 += SGTD_SUB_DESCS) {
    const u32 R = sub_open(Q, B, sg, q, d0, cnt, s_pre[wid], s_ptr[wid], visits);
    total += R;
    // the records of the next four words are loaded while the current four are looked up
    u32 nid[4], ndd[4];
    auto load4 = [&](u32 r0) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const u32 r = r0 + u * SGTD_WAVE + lane;
        u32 ad;
        sub_locate(s_pre[wid], s_ptr[wid], r < R ? r : 0u, ndd[u], ad);
        nid[u] = B.rec[ad];
      }
    };
    if (R) load4(0);
    for (u32 r0 = 0; r0 < R; r0 += 4 * SGTD_WAVE) {
      u32 rid[4], dd[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { rid[u] = nid[u]; dd[u] = ndd[u]; }
      if (r0 + 4 * SGTD_WAVE < R) load4(r0 + 4 * SGTD_WAVE);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool ok = r0 + u * SGTD_WAVE + lane < R;
        u32 sl = 0xFFu;
        const u32 lf = rid[u] >> B.id_bits;                               // local frame
        if (SLOT_TABLE) { if (ok && lf < frame_span) sl = s_slot8[lf]; }   // a dead record's frame is out of range
        else if (ok && lf < frame_span) sl = cand_slot(s_cand, lf + frame_lo);
        const bool valid = sl != 0xFFu;
        const u64 m = __builtin_amdgcn_ballot_w64(valid);
        if (valid) {
          atomicAdd(&s_hist[wid][sl], 1u);   // counting needs no order
          if (fits) {
            const u32 pos = start + running + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            // slot rides in the top 6 bits of the q_idx half (q_idx < 36 * 65535 < 2^26)
            L.pair[pos] = ((u64)((sl << 26) | (d0 + dd[u])) << 32) | (u64)rid[u];
          }
        }
        running += (u32)__builtin_popcountll(m);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  out[lane] = s_hist[wid][lane];
  if (lane == 0) {
    L.blk_start[bslot] = start;
    L.blk_n[bslot] = fits ? running : 0;
    if (q_M) {   // statistics (the key-major pipeline takes them in votes_kernel)
      atomicAdd(&q_M[q], total);
      atomicAdd(&q_P[q], (unsigned long long)visits);
    }
  }
}

// per query: exclusive scan of blk_count over blocks (in place) per slot, then
// over slots: pair_off[q][k] (relative to the query's first pair), q_pairs[q]
__global__ __launch_bounds__(64) void block_scan_kernel(u32 *blk_count, int blocks_per_query, int cand_num,
                                                        const int *n_cand, long long *pair_off, u32 *q_pairs,
                                                        const int *overflow) {
  __shared__ u32 tot[64];
  if (overflow[0]) return;
  const int q = blockIdx.x, s = threadIdx.x;
  u32 *tc = blk_count + (size_t)q * blocks_per_query * 64;
  u32 run = 0;
  int t = 0;
  for (; t + 8 <= blocks_per_query; t += 8) {
    u32 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = tc[(size_t)(t + k) * 64 + s];
#pragma unroll
    for (int k = 0; k < 8; k++) { tc[(size_t)(t + k) * 64 + s] = run; run += v[k]; }
  }
  for (; t < blocks_per_query; t++) { u32 v = tc[(size_t)t * 64 + s]; tc[(size_t)t * 64 + s] = run; run += v; }
  tot[s] = run;
  __syncthreads();
  if (s == 0) {
    const int nc = n_cand[q];
    u32 acc = 0;
    for (int k = 0; k <= cand_num; k++) {
      pair_off[(size_t)q * (cand_num + 1) + k] = acc;
      if (k < nc) acc += tot[k];
    }
    q_pairs[q] = acc;
  }
}

// exclusive scan of q_pairs over queries: q_pair_base[q], [n] = total
__global__ __launch_bounds__(256) void query_base_kernel(const u32 *q_pairs, u32 *q_pair_base, int n_queries,
                                                         u32 pair_cap, int *overflow) {
  __shared__ u32 lds[256 / SGTD_WAVE + 1];
  if (overflow[0]) return;
  u32 carry = 0;
  bool wrapped = false;
  for (int q0 = 0; q0 < n_queries; q0 += 256) {
    const int q = q0 + threadIdx.x;
    const u32 v = (q < n_queries) ? q_pairs[q] : 0;
    u32 tot;
    const u32 ex = block_excl_scan(v, lds, tot);
    if (q < n_queries) q_pair_base[q] = carry + ex;
    if (carry + tot < carry) wrapped = true;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    q_pair_base[n_queries] = carry;
    if (wrapped || carry > pair_cap) overflow[1] = 1;
  }
}

#ifndef SGTD_WRITE_CAP
#define SGTD_WRITE_CAP 16
#endif
// pass 2: every candidate's match_list_ in (i, cell, j) order (:437-449);
// pair = query descriptor index << 32 | insertion index of the table entry
__global__ __launch_bounds__(256) void block_write_kernel(QueryView Q, ProbeBuffers B, CompactLists L,
                                                          int blocks_per_query,
                                                          const u32 *blk_excl, int cand_num,
                                                          const long long *pair_off, const u32 *q_pair_base,
                                                          u64 *pairs, IdMap map) {
  constexpr int NW = 256 / SGTD_WAVE;
  constexpr int CAP = SGTD_WRITE_CAP;   // staged pairs per slot = one 128-B (16) or 64-B (8) line
  __shared__ u64 s_mask[NW][64];         // per wave and slot: lanes of the current word that carry the slot
  __shared__ u64 s_stage[NW][64][CAP + 1];   // per wave and slot: pairs waiting for a full-line store (rows padded by one
                                             // word: a 128-byte row stride put every slot's k-th pair on the same banks)
  if (B.overflow[0] || B.overflow[1]) return;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const BlockId id = assemble_block(Q.n_queries, blocks_per_query);
  if (!id.valid) return;
  const int q = id.q;
  const size_t bslot = (size_t)q * blocks_per_query + id.blk;
  const u32 nv = L.blk_n[bslot];
  if (nv == 0) return;
  const u64 *cp = L.pair + L.blk_start[bslot];
  // lane s carries, for candidate slot s, the output position of its first staged
  // pair (`running`) and the number of staged pairs (`fill`)
  u32 running = 0, fill = 0;
  if (lane < cand_num)
    running = q_pair_base[q] + (u32)pair_off[(size_t)q * (cand_num + 1) + lane] + blk_excl[bslot * 64 + lane];
  // all staged pairs go out as CAP-lane groups, 64 / CAP slots per store instruction
  auto flush = [&]() {
    constexpr int SPI = SGTD_WAVE / CAP;
#pragma unroll 4
    for (int it = 0; it < 64 / SPI; it++) {
      const int s = it * SPI + (lane / CAP), k = lane % CAP;
      const u32 f = __shfl(fill, s), base = __shfl(running, s);
      if ((u32)k < f) pairs[base + k] = s_stage[wid][s][k];
    }
    running += fill;
    fill = 0;
    __builtin_amdgcn_wave_barrier();
  };
  // the next two words of the compact list are loaded while the current two are split (the
  // raw words are only unpacked at the top of the next step: touching them earlier would wait)
  u64 nraw[SGTD_WRITE_WORDS];
  auto load2 = [&](u32 r0) {
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) {
      const u32 r = r0 + u * SGTD_WAVE + lane;
      nraw[u] = cp[r < nv ? r : 0u];
    }
  };
  load2(0);
  for (u32 r0 = 0; r0 < nv; r0 += SGTD_WRITE_WORDS * SGTD_WAVE) {
    u64 pr[SGTD_WRITE_WORDS]; u32 sl[SGTD_WRITE_WORDS];
    u32 first[SGTD_WRITE_WORDS];      // the entry ids become insertion indices here: first entry of the id's frame, fetched
                       // before the next words (loads return in order: it is waited for alone)
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) {
      const bool ok = r0 + u * SGTD_WAVE + lane < nv;
      pr[u] = nraw[u] & 0x03FFFFFFFFFFFFFFull;
      sl[u] = ok ? (u32)(nraw[u] >> 58) : 0xFFu;
      first[u] = map.frame_first[(u32)nraw[u] >> map.bits];
    }
    if (r0 + SGTD_WRITE_WORDS * SGTD_WAVE < nv) load2(r0 + SGTD_WRITE_WORDS * SGTD_WAVE);
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) {
      u32 g = first[u] + ((u32)pr[u] & ((1u << map.bits) - 1u));
      if (map.by_frame) g = map.by_frame[g];
      pr[u] = (pr[u] & 0xFFFFFFFF00000000ull) | (u64)g;
    }
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) {
      const bool valid = sl[u] != 0xFFu;
      const int s = (int)(sl[u] & 63u);
      // lanes with equal slot, by commutative LDS ORs (the result does not depend on the
      // order the hardware applies them in): rank = lanes below me in my group
      s_mask[wid][lane] = 0;
      __builtin_amdgcn_wave_barrier();
      if (valid) atomicOr(&s_mask[wid][s], 1ull << lane);
      __builtin_amdgcn_wave_barrier();
      const u64 gm = valid ? s_mask[wid][s] : 0ull;
      const u32 c_own = (u32)__builtin_popcountll(s_mask[wid][lane]);   // pairs this word adds to slot == lane
      __builtin_amdgcn_wave_barrier();
      const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(gm >> 32), __builtin_amdgcn_mbcnt_lo((u32)gm, 0u)), count = (u32)__builtin_popcountll(gm);
      u32 have = __shfl(fill, s);
      if (__builtin_amdgcn_ballot_w64(valid && have + count > (u32)CAP)) {   // some slot would overflow its line: drain all
        flush();
        have = 0;
      }
      const u32 base = __shfl(running, s);
      const bool direct = count > (u32)CAP;               // a group larger than a line bypasses the stage
      if (valid) {
        if (direct) pairs[base + rank] = pr[u];           // its stage is empty here (just drained)
        else s_stage[wid][s][have + rank] = pr[u];
      }
      if (c_own > (u32)CAP) running += c_own; else fill += c_own;
      __builtin_amdgcn_wave_barrier();
    }
  }
  flush();
}

// diagnostic: the ordered rough-match list of ONE query (reference order i, cell, j).  The
// sweep emits a descriptor's matches segment by segment, cell by cell, inside a cell slice by
// slice; the reference's bucket order is insertion order, so the descriptor's matches are put
// out by ascending (cell, entry id) — tail entries have larger ids than main entries — with a
// selection sort per descriptor (diagnostic path: lists are short).
__global__ __launch_bounds__(256) void rough_gather_kernel(QueryView Q, ProbeBuffers B, IdMap map, int q,
                                                           u32 *out_qi, u32 *out_entry, u32 *out_frame,
                                                           unsigned char *out_cell, double *out_dis) {
  __shared__ u32 lds[256 / SGTD_WAVE + 1];
  const u32 cnt = Q.count[q];
  u32 carry = 0;
  for (u32 i0 = 0; i0 < cnt; i0 += 256) {
    const u32 i = i0 + threadIdx.x;
    const long long d = (long long)q * Q.stride + i;
    u32 n = 0;
    if (i < cnt)
      for (int sg = 0; sg < B.n_seg; sg++) n += B.n_match[(long long)sg * B.seg_stride + d];
    u32 tot;
    const u32 ex = block_excl_scan(n, lds, tot);
    if (i < cnt) {
      long long last = -1;    // key of the last record put out: cell << 32 | entry id
      for (u32 k = 0; k < n; k++) {
        long long best = 0x7FFFFFFFFFFFFFFFll;
        u32 at = 0;
        for (int sg = 0; sg < B.n_seg; sg++) {
          const u32 p0 = B.list_ptr[(long long)sg * B.seg_stride + d], m = B.n_match[(long long)sg * B.seg_stride + d];
          for (u32 j = 0; j < m; j++) {
            const long long key = ((long long)B.rec_cell[p0 + j] << 32) | (long long)id_entry(map, B.rec[p0 + j]);
            if (key > last && key < best) { best = key; at = p0 + j; }
          }
        }
        last = best;
        const u32 o = carry + ex + k;
        out_qi[o] = i;
        out_entry[o] = id_entry(map, B.rec[at]);
        out_frame[o] = id_local_frame(map, B.rec[at]) + map.frame_lo;
        if (out_cell) out_cell[o] = B.rec_cell[at];
        if (out_dis) out_dis[o] = B.rec_dis[at];
      }
    }
    carry += tot;
  }
}
