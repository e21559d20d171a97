// probe_kernels.hip.h — candidate_selector (src/sgtd/src/STDesc.cpp:318-460)
//
//   probe     (:351-400) one wavefront per PASS — up to four query descriptors
//             of one home cell: the 27 cells (truncating (int)(side+inc), gate
//             ||side-centre|| < 1.5, hash lookup key -> bucket) become one
//             concatenated visit list of non-empty ranges — per cell and half of
//             the second side's interval the thirds of the third side's that a
//             descriptor's threshold box reaches, and the bucket's overflow slice
//             (common.hip.h, table_kernels.hip.h); with a tail segment (entries
//             appended after the table was finalized) a cell's main bucket, then
//             its tail bucket, in the same list.  plan_passes_kernel derives the
//             list of every pass, one LANE per pass, and leaves it as a record in
//             HBM; the sweep (probe_sorted_kernel) prefetches the next pass's record
//             while it streams the current list with all 64 lanes from the
//             16-B/entry probe layout (one 16-B load per lane and 64 entries,
//             4 x 64 entries in flight).  The distance test runs in f32 against two
//             squared thresholds that make it conservative on both sides
//             (common.hip.h f32_bounds); the few entries between them are decided on
//             the exact f64 sides with the exact squared threshold (sq_threshold) —
//             every decision is the reference's.  Matches are compacted in visit
//             order by ballot/popcount prefix into a per-descriptor list of 4-byte
//             entry ids (frame and entry in one word, IdMap); restricted to any one
//             map frame that order is the reference's (cell, j) order.
//             Schedule: the batch's descriptors are radix-sorted by home cell,
//             descriptors of one home cell share ONE set of 27 bucket lookups
//             (GroupRow), per-XCD ticket queues hand neighbouring cells to waves of
//             one XCD so that the buckets stay in that XCD's L2; votes_kernel counts
//             votes from the lists afterwards.
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
  u32 whole_at;                // ... and one of more ranges than this even then, with one range per bucket (62; SGTD_WHOLE_AT: test hook)
  // the tail segment (entries appended after the table was finalized), if any: a second directory and
  // hash; its probe layout starts at ent + tail_off (behind the main segment's sentinels), 0: no tail
  u32 tail_off;
  const BucketDir *dir1;
  const HashSlot *hash1;
  u32 hash_mask1;
  u32 n_entries1;
};
// bytes of a home cell's GroupRow slot: the main segment's rows (+ masks) in the first KB, the tail's in the second
__host__ __device__ __forceinline__ u32 group_row_bytes(const TableView &T) { return T.tail_off ? 2u * SGTD_GROUP_ROW_BYTES : SGTD_GROUP_ROW_BYTES; }

struct QueryView {
  const double *side;   // [n_slots*3]
  const QueryRec *qrec; // [n_slots] sweep record of the descriptor (thresholds, gate mask)
  const int *label;     // [n_slots*3]
  const u32 *frame;     // [n_slots]
  const u32 *count;     // [n_queries] descriptors per query
  long long stride;     // descriptor slots per query
  int n_queries;
  u32 chunk;            // query descriptors per block of the block passes (votes, count, write): SGTD_PROBE_CHUNK, or a fraction of it
                        // where a batch has too few blocks to fill the chip (one frame per call: 57 blocks of 128 are 57 WAVES)
};

// A match list starts on a GRANULE of four records (16 bytes) and is named by its granule: 32-bit list starts, slab
// cursors and capacities then address 2^34 records (64 GB of them) — a batch's records outgrow a 32-bit RECORD index
// exactly where large batches pay (2048 query frames on a 100 000-frame map: 1.5e10).  The list passes read records four
// at a time anyway (a quad is a granule: aligned 16-byte loads), and a list wastes at most three records at its end.
#define SGTD_REC_SHIFT 2
struct ProbeBuffers {
  u32 *rec;             // [4 rec_cap] match records: the entry's id (frame and entry in one word, common.hip.h IdMap)
  unsigned char *rec_cell;  // [4 rec_cap] voxel_round index (diagnostic build only)
  double *rec_dis;      // [4 rec_cap] distance (diagnostic build only)
  u32 rec_cap;          // in granules
  u32 rec_slab;         // smallest slab (in granules) a wave takes from the global cursor (SGTD_REC_SLAB records; less for small record buffers:
                        // every stream of every wave holds one, and together they must stay a fraction of the buffer)
  __host__ __device__ __forceinline__ size_t rec_index(u32 granule) const { return (size_t)granule << SGTD_REC_SHIFT; }
  __host__ __device__ __forceinline__ u32 *rec_at(u32 granule) const { return rec + ((size_t)granule << SGTD_REC_SHIFT); }
  u32 rec_rate;         // room a descriptor's list is given when its pass starts: rec_rate / 256 of the visit list (+ 256 records),
                        // at most the whole list; a list that outgrows what its slab has left moves to a new one
  // the batch's counters live in ONE buffer (a single base address in the kernels' scalar
  // registers): words 0-1 the global slab cursor of the match records in granules (64-bit: requests can
  // add up beyond 2^32), 2 the undecided-record queue's fill, 3 the compact lists' cursor,
  // 4-5 matches that found no room (sizes the regrown buffer), 6-7 table entries the sweep
  // really loaded (after slice pruning), 8 the pass pool's cursor, 9 match lists that moved to a fresh
  // slab, 10-11 the overflow flags (0: match records / pass pool / undecided queue, 1: candidate pairs), from word 1024 the
  // sweep's eight ticket-queue heads, 1024 words (4 KB: own L2 channel) apart
  u32 *ctr;
  __host__ __device__ __forceinline__ unsigned long long *rec_cursor() const { return reinterpret_cast<unsigned long long *>(ctr); }
  __host__ __device__ __forceinline__ u32 *amb_count() const { return ctr + 2; }
  __host__ __device__ __forceinline__ u32 *compact_cursor() const { return ctr + 3; }
  __host__ __device__ __forceinline__ unsigned long long *rec_need() const { return reinterpret_cast<unsigned long long *>(ctr + 4); }
  __host__ __device__ __forceinline__ unsigned long long *swept() const { return reinterpret_cast<unsigned long long *>(ctr + 6); }
  __host__ __device__ __forceinline__ u32 *pool_cursor() const { return ctr + 8; }
  __host__ __device__ __forceinline__ u32 *list_moves() const { return ctr + 9; }
  __host__ __device__ __forceinline__ int *overflow() const { return reinterpret_cast<int *>(ctr + 10); }
  __host__ __device__ __forceinline__ u32 *xcd_heads() const { return ctr + 1024; }
  // per descriptor slot d:
  uint2 *list;          // {granule of the first record of the descriptor's list, its matches}
  u32 *n_visit;         // entries the reference's loop visits for the descriptor (STDesc.cpp:372)
  u32 *votes;           // [n_queries * frame_span]
  u32 id_bits;          // a record's local frame (frame - table frame_lo) is rec >> id_bits
  // records whose f32 test fell between the two thresholds: stored provisionally as matches,
  // queued here and decided on the exact sides by resolve_undecided_kernel right after the sweep
  uint2 *amb_queue;     // [amb_cap] (index of the record in the descriptor's list, descriptor slot)
  u32 amb_cap;
};

#define SGTD_PROBE_THREADS 256
#define SGTD_PROBE_CHUNK 128    // query descriptors per assemble block
#define SGTD_REC_SLAB 8192u     // match records a wave takes from the global cursor at once
#define SGTD_SUB_DESCS 32       // descriptors per prefix sub-block inside an assemble block
#ifndef SGTD_VOTE_WORDS
#define SGTD_VOTE_WORDS 6        // 64-quad words of records in flight per wave in the vote pass (votes_topk_kernel, 2048 queries on the
                                 // 10 000-frame map: 2 / 3 / 4 / 6 / 8 words: 1.55-1.62 / 1.52 / 1.49 / 1.39 / 1.49 ms — one 16-wave workgroup
                                 // per CU with 6 KB in flight per wave beats two with 2 KB)
#endif
#ifndef SGTD_WRITE_WORDS
#define SGTD_WRITE_WORDS 4       // 64-pair words of the compact list per step of a wave in block_write (their slot masks take ONE
                                 // LDS round trip together; 2 words: 1.56 ms, 4: 1.51)
#endif

// Loads that were issued before a pass's sweep and are first used after it (the next pass's
// ranges): the sweep "touches" them once its first load group has returned — vector loads
// return in order, so they are complete by then — otherwise the compiler, which cannot count
// the stores of the loops in between, would wait for every outstanding store (vmcnt(0)) at
// their first use.
struct PendingLoads {
  u32 *h = nullptr, *a = nullptr, *b = nullptr, *c = nullptr;
  __device__ __forceinline__ void touch() const {
    if (a) asm volatile("" : "+v"(*h), "+v"(*a), "+v"(*b), "+v"(*c));
  }
};

#ifndef SGTD_PAIR
// descriptors of one home cell swept together by a wave (shared visit list, locate and loads): 1, 2 or 4.
#define SGTD_PAIR 4
#endif
static_assert(SGTD_PAIR == 1 || SGTD_PAIR == 2 || SGTD_PAIR == 4, "the sweep computes the distances two at a time with packed f32 math");
#define SGTD_PASS_KMAX (SGTD_PAIR < 2 ? 2 : SGTD_PAIR)      // descriptor columns of a pass record's header

struct WaveSlab {
  // this wave's private ranges of match records: one bump stream per descriptor column of a pass, so
  // that every descriptor's list stays contiguous.  The streams' state is parked in the lanes of ONE
  // vector register between passes (lane k: next free GRANULE of stream k, lane 4 + k: end of its slab, a granule too;
  // during a pass lane 8 + k: the room, in records, the pass's list has in the slab)
  // — scalar registers are what the sweep runs out of.
  u32 state;
  u64 swept;                             // entries this wave loaded
#ifdef SGTD_EXP_PHASE
  u64 ph[8], t;                          // experiment build: cycles per phase of the wave's life
#endif
};
#ifdef SGTD_EXP_PHASE
// Experiment build (never shipped): in-kernel phase clocks of the sweep, summed over the waves
// and printed by the host after a few launches.
__device__ unsigned long long g_phase[8];
#define PH_ADD(i) do { const u64 _n = __builtin_readcyclecounter(); slab.ph[i] += _n - slab.t; slab.t = _n; } while (0)
#else
#define PH_ADD(i) do { } while (0)
#endif

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

// Below the cell, the home key carries `sub_bits` bits (0..6) of where in the cell's second and third
// interval the descriptor lies (2^(sub_bits / 2) x 2^(sub_bits - sub_bits / 2) classes), so that the
// descriptors of a pass (up to four consecutive positions of one home cell) reach about the same slices of
// the buckets — their shared visit list is the union of what either reaches (4 bits: 14 % fewer
// entries loaded).  The host takes the bits the key leaves free below a multiple of the sort's 8-bit
// digits, or 4 bits and one more sort pass.
// keys/vals are compact: the descriptor (q, i) goes to index q_prefix[q] + i
template <class KeyT>      // u32 where 12 + 3 cbits + sub_bits <= 32 (the shipped resolution), else u64
__global__ void home_keys_kernel(QueryView Q, const u32 *q_prefix, KeyT *keys, u32 *vals, long long n_slots,
                                 int cbits, int sub_bits) {
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
  const int ny = 1 << (sub_bits >> 1), nz = 1 << (sub_bits - (sub_bits >> 1));
  const double f1 = Q.side[d * 3 + 1] - (double)(int)Q.side[d * 3 + 1], f2 = Q.side[d * 3 + 2] - (double)(int)Q.side[d * 3 + 2];
  const u64 sub = (u64)min(max((int)(f1 * ny), 0), ny - 1) * nz + (u64)min(max((int)(f2 * nz), 0), nz - 1);
  keys[idx] = (KeyT)(((((code << cbits | x) << cbits | y) << cbits) | z) << sub_bits | sub);
  vals[idx] = (u32)d;
}

// group id of every sorted position from the SORTED home keys: flags[p] = 1 where the key
// changes or where the next key carries an overflow marker (such a descriptor is a group of
// its own: its key does not name its cell).  flags[p] says whether position p+1 starts a new
// group, so the EXCLUSIVE scan of the flags is the group id of p (position 0 is group 0).
template <class KeyT>
__global__ void group_heads_kernel(const KeyT *keys, const u32 *n_valid_p, u32 *flags, long long n, int cbits, int sub_bits) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  u32 head = 0;
  if (p + 1 < (long long)*n_valid_p) {
    const u64 a = (u64)keys[p + 1] >> sub_bits, b = (u64)keys[p] >> sub_bits;
    const u64 cmask = (1ull << cbits) - 1ull;
    head = (a != b) || ((a & cmask) == cmask) || (((a >> cbits) & cmask) == cmask) ||
           (((a >> (2 * cbits)) & cmask) == cmask);
  }
  flags[p] = head;
}

// first sorted position of every group + the number of groups
__global__ void group_first_kernel(const u32 *gid, const u32 *n_valid_p, u32 *group_first, u32 *n_groups, long long n) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nv = (long long)*n_valid_p;
  if (p >= n || p >= nv) return;
  const u32 g = gid[p];
  if (p == 0 || gid[p - 1] != g) group_first[g] = (u32)p;
  if (p == nv - 1) *n_groups = g + 1;
}

// One GroupRow per home cell of the batch: the 27 ungated bucket lookups (STDesc.cpp:358-371
// minus the gate), each answered with the bucket's directory row {start, cum[0..6]} if the table
// has such a bucket: at rows + g * SGTD_GROUP_ROW_BYTES a quarter of masks (cells with a bucket,
// cells whose bucket has an overflow slice) and the rows of the cells that have one, packed in
// cell order — 10 of 27 on the synthetic maps: the kernel is bound by its traffic (4.8 TB/s with
// all 27 rows written; two or four groups per trip and half-wave, every link of the dependent chain
// issued for all of them first: no faster).
// 32 lanes per group, grid-stride.
__global__ __launch_bounds__(256) void group_resolve_kernel(TableView T, QueryView Q, const u32 *order,
                                                            const u32 *group_first, const u32 *n_groups_p,
                                                            const u32 *n_valid_p, unsigned char *rows, u32 rows_cap,
                                                            int *overflow) {
  const int c = (int)(threadIdx.x & 31);
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 5;
  const long long n_groups = (*n_valid_p) ? (long long)*n_groups_p : 0;
  if (n_groups > (long long)rows_cap) {     // more home cells than GroupRows were reserved: the host reserves more and re-runs the batch
    if (blockIdx.x == 0 && threadIdx.x == 0) overflow[0] = 1;
    return;
  }
  for (long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5; g < n_groups; g += stride) {
    if (c >= SGTD_NCELL) continue;      // (the ballots below only look at lanes c < 27 of either half)
    const long long d = (long long)order[group_first[g]];
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
    // a side so small that side - 1 rounds to -1.0 (|side| below 2^-53): the reference probes the
    // empty cell -1, the clamp above would sweep cell 0 a second time — that slot stays empty
    const bool minus_one = nonneg && ((ix < 0 && (int)(q0 + (double)ix) < 0) || (iy < 0 && (int)(q1 + (double)iy) < 0) ||
                                      (iz < 0 && (int)(q2 + (double)iz) < 0));
    const bool probe = !minus_one && x >= 0 && y >= 0 && z >= 0 && x < 65536 && y < 65536 && z < 65536;
    const u64 key = probe ? pack_key(code, (u32)x, (u32)y, (u32)z) : 0ull;
    for (int sg = 0; sg < (T.tail_off ? 2 : 1); sg++) {     // main segment, tail segment
      const HashSlot *hash = sg ? T.hash1 : T.hash;
      const BucketDir *dir = sg ? T.dir1 : T.dir;
      const u32 mask = sg ? T.hash_mask1 : T.hash_mask;
      uint4 lo = make_uint4(0, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
      if (probe) {
        u32 h = hash_key(key) & mask;
        while (true) {
          const HashSlot s = hash[h];
          if (s.key == key) {
            const uint4 *row = reinterpret_cast<const uint4 *>(dir + s.bucket);
            lo = row[0]; hi = row[1];
            break;
          }
          if (s.key == SGTD_EMPTY_KEY) break;
          h = (h + 1) & mask;
        }
      }
      // the set's first quarter: which cells have a bucket at all, and which of those an overflow slice
      // (plan_passes_kernel bounds a pass's number of ranges with them before it walks the rows); behind it
      // the rows of the cells that have one, packed in cell order (two quarters each)
      const u64 ex = __builtin_amdgcn_ballot_w64(hi.w != 0u), ov = __builtin_amdgcn_ballot_w64(hi.w != hi.z);
      const int sh = (int)(threadIdx.x & 32);
      const u32 exh = (u32)(ex >> sh) & 0x7FFFFFFu;
      uint4 *out = reinterpret_cast<uint4 *>(rows + (size_t)g * group_row_bytes(T) + (size_t)sg * SGTD_GROUP_ROW_BYTES);
      if (hi.w != 0u) {
        const u32 r = 1u + 2u * (u32)__builtin_popcount(exh & ((1u << c) - 1u));
        out[r] = lo;         // start, cum[0..2]
        out[r + 1] = hi;     // cum[3..6]
      }
      if (c == 0) out[0] = make_uint4(exh, (u32)(ov >> sh) & 0x7FFFFFFu, 0u, 0u);
    }
  }
}

// ---------------------------------------------------------------------------
// Passes.  A PASS is what one wavefront sweeps at a time: up to SGTD_PAIR (four) consecutive
// descriptors of one home cell (sorted positions first + 4 i .. first + 4 i + 3 of the group —
// they share the GroupRow, so one visit list, one locate and one 16-B load per 64 entries serve
// all of them).  plan_passes_kernel turns every pass into a RECORD in HBM, one lane per
// pass, so that the sweep starts from a ready visit list instead of deriving it per wave:
//
//   header, 2 + 7 KM words for KM = SGTD_PASS_KMAX descriptor columns (64 B for 2, 128 B for 4: one 4-B load
//   per lane, in flight during the pass before; v_readlane at the pass's start):
//     w0 = n | K << 8 | R << 12   n = non-empty ranges of the visit list (<= 62), K = descriptor columns
//                         the sweep tests (1, 2, 4), R = descriptors of the pass (a pass of three is swept
//                         with a fourth column whose gate fails everywhere)
//     w1 = total          entries in the visit list
//     then KM words each (pass_hdr_word): descriptor slots d; the descriptors' frames as the entry ids name
//     them (frame - frame_lo, 0xFFFFFFFF for a frame the table does not hold); q0, q1, q2 (f32); lo2, hi2
//     (f32_bounds)
//   ranges, 12 B each, n + 1 of them (lane j of the sweep loads range j):
//     offc   exclusive offset of range j in the visit list
//     dlc    start - offc (entry index of list position p in range j = p + dlc)
//     meta   cell (0..26) | bit 8 + k: the cell fails descriptor k's gate
//   Range n is the sentinel: it starts at `total` and maps onto the 64 entries with sides +inf
//   behind the table's last entry, so the lanes of the last 64-entry word beyond the list need
//   no special case.
//
// The visit list: the non-empty ranges of table entries in the order of the reference's cells
// (voxel_round, :327-333; inside a cell: lower half, upper half, overflow slice) — per cell
// and half of the second side's interval the thirds of the third side's that hold entries with
// |side1 - q1| <= thr' and |side2 - q2| <= thr' (an entry outside cannot match: its squared
// distance, as the reference computes it, is at least fl(d * d) >= thr2 for that axis), and the
// bucket's overflow slice (all of it; most buckets have none).  For a pass of several descriptors
// the list is the union of their lists; the sweep tests every loaded entry against each of them and
// starts each sum from the descriptor's own gate penalty (a slice too many is harmless, a cell
// too many is not).
//
// Pass slots: leader position p = f + KM i of group g (first position f) owns slot
// g + ceil(f / KM) + i — injective and increasing in p (ceil(a) + ceil(b) <= ceil(a + b) + 1), at most
// n_groups + ceil(n_valid / KM) + 1 slots, the unused ones (at most one per group) hold
// SGTD_NO_PASS; the sweep's tickets are ranges of slots.
// ---------------------------------------------------------------------------
#define SGTD_NO_PASS 0xFFFFFFFFu
#define SGTD_PASS_HDR_WORDS (SGTD_PASS_KMAX == 4 ? 32 : 16)
#define SGTD_PASS_HDR_UNITS (SGTD_PASS_HDR_WORDS / 4)          // the header in 16-B units
#define SGTD_META_SENTINEL (((1u << SGTD_PASS_KMAX) - 1u) << 8)   // every gate fails
enum { PH_SLOT = 0, PH_FRAME, PH_Q0, PH_Q1, PH_Q2, PH_LO2, PH_HI2 };
__host__ __device__ constexpr int pass_hdr_word(int field, int k) { return 2 + field * SGTD_PASS_KMAX + k; }
#define SGTD_PASS_SLACK_UNITS 64       // behind the pool: the sweep's 64 lanes load 12 B each whatever n is

struct __attribute__((packed, aligned(4))) RangeWords { u32 offc, dlc, meta; };

struct PassPool {
  uint4 *pool;          // [cap + slack] pass records, 16-B units
  u32 *rec_off;         // [pass slots] first unit of the slot's record, SGTD_NO_PASS: nothing to sweep
  u32 *cursor;          // units handed out (keeps counting when the pool is full: sizes the regrown pool)
  u32 cap;              // units
};

__host__ __device__ __forceinline__ u32 pass_slot_count(u32 n_valid, u32 n_groups, bool pair) {
  if (!n_valid) return 0u;
  return pair ? n_groups + (n_valid + SGTD_PAIR - 1u) / SGTD_PAIR + 1u : n_valid;
}

// pos_of_slot[s] = sorted position of the leader of pass slot s (the array is pre-set to SGTD_NO_PASS)
__global__ void pass_slots_kernel(const u32 *gid, const u32 *group_first, const u32 *n_valid_p, u32 *pos_of_slot,
                                  long long n, int pair) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || p >= (long long)*n_valid_p) return;
  if (!pair) { pos_of_slot[p] = (u32)p; return; }
  const u32 g = gid[p], f = group_first[g];
  if ((((u32)p - f) % SGTD_PAIR) == 0u) pos_of_slot[g + (f + SGTD_PAIR - 1u) / SGTD_PAIR + ((u32)p - f) / SGTD_PAIR] = (u32)p;
}

// ---- ONE frame per call (nq = 1, at most 8192 descriptor slots, 32-bit home keys): everything between the descriptors and
// the GroupRows — the batch's counters and result tables cleared, query_prefix, home_keys, the four radix passes, group_heads, the
// scan, group_first, pos_of_slot's reset and pass_slots: twenty-seven launches of about 5 us each on the device's timeline —
// by ONE workgroup in LDS (a stable radix sort of (home key << 13 | slot) by the key's digits).
#define SGTD_SMALL_SLOTS 8192
#define SGTD_SMALL_THREADS 1024
struct SmallOrder {
  u32 *ctr; u32 ctr_words;                  // ProbeBuffers::ctr, cleared
  u32 *q_M; unsigned long long *q_P;        // [1], cleared
  u32 *votes; u32 *slot_of_words; u32 span; // block passes: the vote histogram (zero) and the frame -> slot bytes (0xFF), span frames
  int *cand_frame, *cand_votes; int cand_num;
  u32 *q_prefix, *n_valid, *order, *gid, *group_first, *n_groups, *pos_of_slot;
  u32 n_slots, max_pass_slots;
  int cbits, sub_bits, pair, key_bits;
  QueryRec *qrec; u32 n_qrec; double rough;  // descriptors handed in by the caller: their sweep records are written here too (thr2_kernel)
};
__global__ __launch_bounds__(SGTD_SMALL_THREADS) void small_order_kernel(QueryView Q, SmallOrder S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  u64 *s_key = reinterpret_cast<u64 *>(s_raw);                                 // [8192] home key << 13 | slot
  u32 *s_gid = reinterpret_cast<u32 *>(s_raw + SGTD_SMALL_SLOTS * 8);          // [8192]
  u32 *s_first = s_gid + SGTD_SMALL_SLOTS;                                     // [8192]
  __shared__ u32 s_sum[SGTD_SMALL_THREADS / SGTD_WAVE + 1];
  __shared__ u32 s_cnt[4096];                                                  // the sort's counts per (digit, wave)
  const u32 tid = threadIdx.x;
  // ---- what the memsets did
  for (u32 i = tid; i < S.ctr_words; i += SGTD_SMALL_THREADS) S.ctr[i] = 0u;
  if (tid == 0) { S.q_M[0] = 0u; S.q_P[0] = 0ull; S.q_prefix[0] = 0u; }
  if (S.votes) for (u32 i = tid; i < S.span; i += SGTD_SMALL_THREADS) S.votes[i] = 0u;
  if (S.slot_of_words) for (u32 i = tid; i < (S.span + 3u) / 4u; i += SGTD_SMALL_THREADS) S.slot_of_words[i] = 0xFFFFFFFFu;
  if (S.cand_frame) for (u32 i = tid; i < (u32)S.cand_num; i += SGTD_SMALL_THREADS) { S.cand_frame[i] = -1; S.cand_votes[i] = 0; }
  for (u32 i = tid; i < S.max_pass_slots; i += SGTD_SMALL_THREADS) S.pos_of_slot[i] = SGTD_NO_PASS;
  if (S.qrec)
    for (u32 i = tid; i < S.n_qrec; i += SGTD_SMALL_THREADS) write_query_rec(S.qrec + i, Q.side[i * 3], Q.side[i * 3 + 1], Q.side[i * 3 + 2], S.rough, Q.frame[i]);
  // ---- home keys (home_keys_kernel<u32>)
  const u32 nv = min(Q.count[0], S.n_slots);
  u32 P = SGTD_SMALL_THREADS;                    // (a multiple of 64 positions for every wave)
  while (P < nv) P <<= 1;
  const u64 cmask = (1ull << S.cbits) - 1ull;
  const int ny = 1 << (S.sub_bits >> 1), nz = 1 << (S.sub_bits - (S.sub_bits >> 1));
  for (u32 d = tid; d < P; d += SGTD_SMALL_THREADS) {
    u64 k = ~0ull;
    if (d < nv) {
      const u64 code = label_code(Q.label[d * 3 + 0], Q.label[d * 3 + 1], Q.label[d * 3 + 2]);
      const u64 x = min((u64)(u32)(int)Q.side[d * 3 + 0], cmask), y = min((u64)(u32)(int)Q.side[d * 3 + 1], cmask),
                z = min((u64)(u32)(int)Q.side[d * 3 + 2], cmask);
      const double f1 = Q.side[d * 3 + 1] - (double)(int)Q.side[d * 3 + 1], f2 = Q.side[d * 3 + 2] - (double)(int)Q.side[d * 3 + 2];
      const u64 sub = (u64)min(max((int)(f1 * ny), 0), ny - 1) * nz + (u64)min(max((int)(f2 * nz), 0), nz - 1);
      const u32 key = (u32)(((((code << S.cbits | x) << S.cbits | y) << S.cbits) | z) << S.sub_bits | sub);
      k = ((u64)key << 13) | (u64)d;
    }
    s_key[d] = k;
  }
  __syncthreads();
  // ---- stable LSD radix sort of the P keys by their home key (8-bit digits, the passes the key has bits for): wave w owns the
  // P / 16 consecutive positions [w P / 16, (w + 1) P / 16); counts per (digit, wave) in LDS, one scan over them in (digit, wave)
  // order, then every wave places its positions chunk by chunk (rank inside a chunk: eight ballots, wave_group_rank)
  {
    u64 *src = s_key, *dst = reinterpret_cast<u64 *>(s_raw + SGTD_SMALL_SLOTS * 8);
    const u32 wid = tid >> 6, lane = tid & 63u, per_wave = P / (SGTD_SMALL_THREADS / SGTD_WAVE);
    for (int sh = 0; sh < S.key_bits; sh += 8) {
      for (u32 i = tid; i < 4096u; i += SGTD_SMALL_THREADS) s_cnt[i] = 0u;
      __syncthreads();
      for (u32 c = 0; c < per_wave; c += SGTD_WAVE) {      // (a chunk's equal digits as ONE add by their first lane: sixty-four atomic adds to a
        const u32 d = (u32)((src[wid * per_wave + c + lane] >> (13 + sh)) & 255ull);      // word that most of the chunk shares take each other's turn)
        u32 rank, count;
        wave_group_rank<8>(d, true, rank, count);
        const u32 have = s_cnt[d * 16u + wid];
        __builtin_amdgcn_wave_barrier();
        if (rank == 0) s_cnt[d * 16u + wid] = have + count;
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      {
        u32 v[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { v[j] = s_cnt[tid * 4 + j]; sum += v[j]; }
        u32 tot;
        u32 ex = block_excl_scan(sum, s_sum, tot);
#pragma unroll
        for (int j = 0; j < 4; j++) { s_cnt[tid * 4 + j] = ex; ex += v[j]; }
      }
      __syncthreads();
      for (u32 c = 0; c < per_wave; c += SGTD_WAVE) {
        const u64 k = src[wid * per_wave + c + lane];
        const u32 d = (u32)((k >> (13 + sh)) & 255ull);
        u32 rank, count;
        wave_group_rank<8>(d, true, rank, count);
        const u32 at = s_cnt[d * 16u + wid];
        __builtin_amdgcn_wave_barrier();
        if (rank == 0) s_cnt[d * 16u + wid] = at + count;
        dst[at + rank] = k;
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      u64 *t = src; src = dst; dst = t;
    }
    if (src != s_key) {
      for (u32 i = tid; i < P; i += SGTD_SMALL_THREADS) s_key[i] = src[i];
      __syncthreads();
    }
  }
  // ---- group ids: flags[p] = position p + 1 starts a new group (group_heads_kernel), gid = their exclusive scan
  constexpr u32 PER = SGTD_SMALL_SLOTS / SGTD_SMALL_THREADS;      // 8 consecutive positions per thread
  u32 fl[PER], mine = 0;
#pragma unroll
  for (u32 j = 0; j < PER; j++) {
    const u32 p = tid * PER + j;
    u32 head = 0;
    if (p + 1 < nv) {
      const u64 a = (s_key[p + 1] >> 13) >> S.sub_bits, b = (s_key[p] >> 13) >> S.sub_bits;
      head = (a != b) || ((a & cmask) == cmask) || (((a >> S.cbits) & cmask) == cmask) || (((a >> (2 * S.cbits)) & cmask) == cmask);
    }
    fl[j] = head;
    mine += head;
  }
  u32 tot;
  u32 ex = block_excl_scan(mine, s_sum, tot);
#pragma unroll
  for (u32 j = 0; j < PER; j++) {
    const u32 p = tid * PER + j;
    s_gid[p] = ex;
    if (p < S.n_slots) { S.gid[p] = ex; if (p < nv) S.order[p] = (u32)(s_key[p] & 8191ull); }
    ex += fl[j];
  }
  __syncthreads();
  // ---- first position of every group, the number of groups
  for (u32 p = tid; p < nv; p += SGTD_SMALL_THREADS) {
    const u32 g = s_gid[p];
    if (p == 0 || s_gid[p - 1] != g) { s_first[g] = p; S.group_first[g] = p; }
  }
  if (tid == 0) { *S.n_valid = nv; *S.n_groups = nv ? s_gid[nv - 1] + 1u : 0u; }
  __syncthreads();
  // ---- pass slots (pass_slots_kernel)
  for (u32 p = tid; p < nv; p += SGTD_SMALL_THREADS) {
    if (!S.pair) { S.pos_of_slot[p] = p; continue; }
    const u32 g = s_gid[p], f = s_first[g];
    if (((p - f) % SGTD_PAIR) == 0u) S.pos_of_slot[g + (f + SGTD_PAIR - 1u) / SGTD_PAIR + (p - f) / SGTD_PAIR] = p;
  }
}

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

// One lane per pass slot.  The lanes of a wave hold consecutive slots, i.e. passes of consecutive
// groups.  Per trip of a workgroup (256 slots):
//   1. every lane reads its pass (positions, descriptors, its group's masks: which of the 27 cells have a
//      bucket / an overflow slice — group_resolve_kernel left them behind the rows — and the descriptors'
//      records) and bounds its record: two ranges per gated cell that has a bucket, one more where the
//      bucket has an overflow slice;
//   2. room for all 256 records with ONE add to the pool's cursor (a scan per wave, the waves' sums through
//      LDS);
//   3. wave by wave, SGTD_PLAN_GROUPS groups at a time: the rows of the cells that HAVE a bucket (10 of 27
//      on the synthetic maps) are staged in LDS, packed, behind a zero row that stands for all others (a
//      set = one group's rows in one segment, packed like that by group_resolve_kernel: a flat copy; groups that do not
//      fit the wave's 7 KB wait for the next round), and the lanes of those groups walk THEIR cells there
//      — gated by one of their descriptors and holding a bucket — writing the ranges as they go.
// A pass without a single entry to visit gets no record and its (empty) results are written here.
#ifndef SGTD_PLAN_GROUPS
#define SGTD_PLAN_GROUPS 16   // (4 / 6 / 8 / 12 groups per round measured +1.07 / +0.50 / +0.24 / +0 ms: a round costs a whole walk)
#endif
#define SGTD_PLAN_THREADS 256
#ifndef SGTD_PLAN_WAVES
#define SGTD_PLAN_WAVES __attribute__((amdgpu_waves_per_eu(5)))
#endif
#ifndef SGTD_EXP_PLAN_STAGE
// Experiment builds (never shipped): the planner stops behind stage 1 .. 5 (slot -> positions and descriptors -> the
// group's masks -> the descriptors' records and reach -> room) or, 6, does everything but the walk; with the product
// build, the differences are what each stage costs.
#define SGTD_EXP_PLAN_STAGE 0
#endif
#ifndef SGTD_PLAN_INFLIGHT
#define SGTD_PLAN_INFLIGHT 4u   // sets whose rows are in flight together in the staging copy
#endif
#ifndef SGTD_PLAN_SPARSE
#define SGTD_PLAN_SPARSE 1
#endif
#ifndef SGTD_PLAN_QUADS
#define SGTD_PLAN_QUADS 448    // staging room per wave, 16-B quarters: the rows of the cells that HAVE a bucket (10 of a group's 27 on
                               // the synthetic maps) of up to SGTD_PLAN_GROUPS groups; groups that do not fit wait for the next round
#endif
#define SGTD_ROW_QUADS (2 * SGTD_NCELL + 1)     // 16-B quarters of one GroupRow: 27 rows + the masks
// TAIL: the table has a tail segment — a GroupRow slot holds two sets of rows, half as many groups are
// staged per round, and a cell's ranges are the main segment's followed by the tail's (the reference's
// bucket holds the appended entries behind the older ones).
template <bool PAIR, bool TAIL>
__global__ __launch_bounds__(SGTD_PLAN_THREADS) SGTD_PLAN_WAVES void plan_passes_kernel(TableView T, QueryView Q, const u32 *order, const u32 *gid,
                                                          const u32 *pos_of_slot, const u32 *n_valid_p,
                                                          const u32 *n_groups_p, const unsigned char *rows, u32 rows_cap, PassPool P,
                                                          u32 *n_visit, uint2 *list, int *overflow) {
  constexpr u32 SEGS = TAIL ? 2u : 1u;
  constexpr u32 NG = SGTD_PLAN_GROUPS / SEGS;                 // groups staged per round
  constexpr size_t ROWB = (size_t)SEGS * SGTD_GROUP_ROW_BYTES;
  // staged per wave: a zero row (the cells without a bucket) + the rows of the cells that have one, set after set
  // (a set = one group's rows in one segment)
  __shared__ uint4 s_rows[SGTD_PLAN_THREADS / SGTD_WAVE][SGTD_PLAN_QUADS + 2];
  const int lane = lane_id();
  uint4 *my_rows = s_rows[threadIdx.x >> 6];
  if (lane < 2) my_rows[lane] = make_uint4(0u, 0u, 0u, 0u);
  __builtin_amdgcn_wave_barrier();
  const u32 nv = *n_valid_p;
  const u32 n_pass = pass_slot_count(nv, *n_groups_p, PAIR);
  const bool no_rows = *n_groups_p > rows_cap;      // group_resolve_kernel raised the overflow flag: no pass gets a record
  // every wave takes 64 consecutive slots at a time, the workgroup SGTD_PLAN_THREADS, grid-stride (the grid is sized by
  // resident workgroups); the trip count is the same for all waves of a workgroup (barriers below)
  __shared__ u32 s_units[SGTD_PLAN_THREADS / SGTD_WAVE], s_base[SGTD_PLAN_THREADS / SGTD_WAVE], s_room;
  for (u32 s_wg = blockIdx.x * blockDim.x; s_wg < n_pass; s_wg += gridDim.x * blockDim.x) {
  const u32 s = s_wg + threadIdx.x;
  const u32 p = (s < n_pass && !no_rows) ? pos_of_slot[s] : SGTD_NO_PASS;
#if SGTD_EXP_PLAN_STAGE == 1
  if (p == 0x12345678u) overflow[1] = 1;
  if (s < n_pass) P.rec_off[s] = SGTD_NO_PASS;
  continue;
#endif
  constexpr int KM = SGTD_PASS_KMAX;
  int K = 0;
  u32 g = 0, d[KM];
#pragma unroll
  for (int k = 0; k < KM; k++) d[k] = 0;
  bool act = p != SGTD_NO_PASS;
  u32 ex_g[SEGS], ov_g[SEGS];     // the group's masks: cells that have a bucket / an overflow slice, per segment
#pragma unroll
  for (u32 sg = 0; sg < SEGS; sg++) { ex_g[sg] = 0; ov_g[sg] = 0; }
  if (act) {
    g = gid[p];
    K = 1;
    if (PAIR) {
#pragma unroll
      for (int k = 1; k < SGTD_PAIR; k++) K += (K == k && p + k < nv && gid[p + k] == g) ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < KM; k++) d[k] = order[p + (k < K ? k : 0)];
    // a home cell none of whose 27 buckets exists in the table: nothing to plan or sweep — before
    // anything of the descriptors is read
    u32 any = SGTD_EXP_PLAN_STAGE == 2 ? 1u : 0u;
#pragma unroll
    for (u32 sg = 0; sg < (SGTD_EXP_PLAN_STAGE == 2 ? 0u : SEGS); sg++) {
      const uint2 m = *reinterpret_cast<const uint2 *>(rows + (size_t)g * ROWB + (size_t)sg * SGTD_GROUP_ROW_BYTES);
      ex_g[sg] = m.x; ov_g[sg] = m.y;
      any |= m.x;
    }
    if (any == 0u) {
#pragma unroll
      for (int k = 0; k < KM; k++)
        if (k < K) { n_visit[d[k]] = 0; list[d[k]] = make_uint2(0u, 0u); }
      act = false;
    }
  }
  u32 hq[KM][5];   // q0, q1, q2 (f32), lo2, hi2 as words
#if SGTD_EXP_PLAN_STAGE == 3 || SGTD_EXP_PLAN_STAGE == 2
  if ((ex_g[0] ^ d[0] ^ d[1] ^ d[2] ^ d[3]) == 0x12345678u) overflow[1] = 1;
  if (s < n_pass) P.rec_off[s] = SGTD_NO_PASS;
  continue;
#endif
  if (!act) K = 0;
  u32 qfr[KM], gate[KM];
  // per descriptor and offset -1, 0, +1: the halves (second side) and thirds (third side) its
  // threshold box reaches as bit masks: halves in bits 2 o .. 2 o + 1, thirds in bits 8 + 3 o .. 8 + 3 o + 2
  u32 reach[KM];
#pragma unroll
  for (int k = 0; k < KM; k++) {
    qfr[k] = 0xFFFFFFFFu; gate[k] = 0; reach[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) hq[k][i] = 0;
    if (k < K) {
      const uint4 *r = reinterpret_cast<const uint4 *>(Q.qrec + d[k]);
      const uint4 a = r[0], b = r[1], c = r[2], e = r[3];
      const double q0 = __hiloint2double((int)a.y, (int)a.x), q1 = __hiloint2double((int)a.w, (int)a.z),
                   q2 = __hiloint2double((int)b.y, (int)b.x);
      hq[k][0] = __float_as_uint((float)q0); hq[k][1] = __float_as_uint((float)q1); hq[k][2] = __float_as_uint((float)q2);
      hq[k][3] = e.x; hq[k][4] = e.y;
      const u32 ql = c.x - T.map.frame_lo;
      qfr[k] = ql < T.frame_span ? ql : 0xFFFFFFFFu;
      gate[k] = c.y;
      const float t_up = __uint_as_float(e.z);
#pragma unroll
      for (int o = 0; o < 3; o++) {
        int lo, hi;
        reached_slices((float)(q1 - (double)(int)(q1 + (double)(o - 1))), t_up, SGTD_YSLICES, lo, hi);
        reach[k] |= (((1u << (hi + 1)) - 1u) & ~((1u << lo) - 1u)) << (2 * o);           // (hi < lo: no bit)
        reached_slices((float)(q2 - (double)(int)(q2 + (double)(o - 1))), t_up, SGTD_ZSLICES, lo, hi);
        reach[k] |= (((1u << (hi + 1)) - 1u) & ~((1u << lo) - 1u)) << (8 + 3 * o);
      }
    }
  }
  u32 gate_any = 0;
#pragma unroll
  for (int k = 0; k < KM; k++) gate_any |= gate[k];
#if SGTD_EXP_PLAN_STAGE == 4
  { u32 x = gate_any; for (int k = 0; k < KM; k++) x ^= reach[k] ^ hq[k][0] ^ hq[k][4] ^ qfr[k]; if (x == 0x12345678u) overflow[1] = 1; }
  if (s < n_pass) P.rec_off[s] = SGTD_NO_PASS;
  continue;
#endif
  // Room for the records, for all 64 x waves passes of the workgroup at once: an upper bound of a pass's ranges — two
  // per gated cell that has a bucket, one more where the bucket has an overflow slice — from the masks alone, one
  // scan per wave and ONE add to the pool's cursor per workgroup and trip (an address takes 88 M atomic adds per
  // second, tools/atomic_rate.hip: one add per wave and staging round, 8 x 10^4 a batch, were 0.9 of this kernel's
  // 1.1 ms)
  u32 n_ex = 0, n_ov = 0;
#pragma unroll
  for (u32 sg = 0; sg < SEGS; sg++) {
    n_ex += (u32)__builtin_popcount(gate_any & ex_g[sg]);
    n_ov += (u32)__builtin_popcount(gate_any & ov_g[sg]);
  }
  u32 ub = act ? 2u * n_ex + n_ov : 0u;
  // more ranges than the sweep has lanes (only with many overflow slices): the halves of a cell as
  // ONE unpruned range, all six sub-cells — a superset of what the descriptors reach; with a tail
  // segment even that can be too many: then every bucket is one range, overflow slice included (it
  // follows the halves)
  const bool coarse = ub > T.coarse_at;
  if (coarse) ub = n_ex + n_ov;
  const bool whole = TAIL && coarse && ub > T.whole_at;
  if (whole) ub = n_ex;
  const u32 units = ub ? SGTD_PASS_HDR_UNITS + ((ub + 1u) * 12u + 15u) / 16u : 0u;
  const u32 inc = wave_incl_scan(units);
  if (lane == SGTD_WAVE - 1) s_units[threadIdx.x >> 6] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 tot = 0;
#pragma unroll
    for (int w = 0; w < SGTD_PLAN_THREADS / SGTD_WAVE; w++) { s_base[w] = tot; tot += s_units[w]; }
    const u32 base = tot ? atomicAdd(P.cursor, tot) : 0u;
#pragma unroll
    for (int w = 0; w < SGTD_PLAN_THREADS / SGTD_WAVE; w++) s_base[w] += base;
    const bool fits = (u64)base + tot <= (u64)P.cap;
    s_room = fits ? 1u : 0u;
    if (!fits) overflow[0] = 1;                 // the host grows the pool and re-runs the batch
  }
  __syncthreads();
  const bool room = s_room != 0u;
  const u32 off = s_base[threadIdx.x >> 6] + inc - units;
#if SGTD_EXP_PLAN_STAGE == 5
  { u32 x = gate_any ^ off; for (int k = 0; k < KM; k++) x ^= reach[k] ^ hq[k][0] ^ hq[k][4] ^ qfr[k]; if (x == 0x12345678u) overflow[1] = 1; }
  if (s < n_pass) P.rec_off[s] = SGTD_NO_PASS;
  continue;
#endif
  // the wave's groups are consecutive ids (slots grow with the sorted position): first and last active lane
  const u64 act_mask = __builtin_amdgcn_ballot_w64(act);
  if (!act_mask) {
    if (s < n_pass) P.rec_off[s] = SGTD_NO_PASS;
    continue;
  }
  const u32 g_lo = (u32)__builtin_amdgcn_readlane((int)g, __builtin_ctzll(act_mask));
  const u32 g_hi = (u32)__builtin_amdgcn_readlane((int)g, 63 - __builtin_clzll(act_mask));
  u32 my_off = SGTD_NO_PASS;
  for (u32 gc = g_lo, ng = 0; gc <= g_hi; gc += ng) {
    // the masks of the next (up to) NG groups' sets, one lane each; then as many whole groups as the staging room takes
    const u32 ng_max = min(NG, g_hi - gc + 1u);
    u32 ex_l = 0;
    if ((u32)lane < ng_max * SEGS) {
      const u32 gi = gc + (u32)lane / SEGS, sg = (u32)lane % SEGS;
      ex_l = *reinterpret_cast<const u32 *>(rows + (size_t)gi * ROWB + (size_t)sg * SGTD_GROUP_ROW_BYTES);
    }
    const u32 sz = 2u * (u32)__builtin_popcount(ex_l);            // quarters of the set's rows
    const u32 incl = wave_incl_scan(sz);
    const u32 pre_l = incl - sz + 2u;                              // its first quarter (behind the zero row)
    {
      const u64 fitm = __builtin_amdgcn_ballot_w64(incl <= (u32)SGTD_PLAN_QUADS);      // (leading ones: incl grows with the lane)
      const u32 n_fit = ~fitm ? (u32)__builtin_ctzll(~fitm) : 64u;
      ng = min(n_fit, ng_max * SEGS) / SEGS;                       // (at least one: a group's sets are 54 SEGS quarters)
    }
    __builtin_amdgcn_wave_barrier();
    // set after set: lane q copies quarter q of the set's packed rows; SGTD_PLAN_INFLIGHT sets' loads in flight
    for (u32 s0 = 0; s0 < ng * SEGS; s0 += SGTD_PLAN_INFLIGHT) {
      uint4 t[SGTD_PLAN_INFLIGHT];
      u32 dst[SGTD_PLAN_INFLIGHT];
#pragma unroll
      for (u32 u = 0; u < SGTD_PLAN_INFLIGHT; u++) {
        const u32 st = min(s0 + u, ng * SEGS - 1u);
        const u32 sz_s = (u32)__builtin_amdgcn_readlane((int)sz, (int)st), pre_s = (u32)__builtin_amdgcn_readlane((int)pre_l, (int)st);
        const bool on = s0 + u < ng * SEGS && (u32)lane < sz_s;
        dst[u] = on ? pre_s + (u32)lane : 0xFFFFFFFFu;
        // (lanes beyond the set's rows read its mask quarter: one line for all of them, nothing stored)
        t[u] = reinterpret_cast<const uint4 *>(rows + (size_t)(gc + st / SEGS) * ROWB + (size_t)(st % SEGS) * SGTD_GROUP_ROW_BYTES)[on ? 1 + lane : 0];
      }
#pragma unroll
      for (u32 u = 0; u < SGTD_PLAN_INFLIGHT; u++)
        if (dst[u] != 0xFFFFFFFFu) my_rows[dst[u]] = t[u];
    }
    __builtin_amdgcn_wave_barrier();
    const bool mine = act && g - gc < ng;      // (g >= gc for every lane not yet served)
    // the first quarters of the lane's sets, from the lanes that hold them
    u32 pre_m[SEGS], rk[SEGS];
#pragma unroll
    for (u32 sg = 0; sg < SEGS; sg++) {
      pre_m[sg] = (u32)__builtin_amdgcn_ds_bpermute((int)(((mine ? g - gc : 0u) * SEGS + sg) << 2), (int)pre_l);
      rk[sg] = 0;
    }
    u32 n = 0, total = 0, visits[KM];
#pragma unroll
    for (int k = 0; k < KM; k++) visits[k] = 0;
    if (mine) {
      RangeWords *wp = reinterpret_cast<RangeWords *>(P.pool + off + SGTD_PASS_HDR_UNITS);   // next range of the record
      const bool emit = ub != 0u && room;
      auto put = [&](u32 start, u32 len, u32 meta) {
        if (len) {
          if (emit) *wp++ = RangeWords{total, start - total, meta};     // one 12-B store
          total += len;
          n++;
        }
      };
      // one cell of one segment: its ranges, its visits
      auto cell = [&](int c, u32 sgi, u32 rank) {
        const int oy = (c / 3) % 3, oz = c % 3;
        const u32 has = (ex_g[sgi] >> c) & 1u;                                                 // (no bucket: the zero row)
        const u32 *rc = reinterpret_cast<const u32 *>(my_rows + (has ? pre_m[sgi] + 2u * rank : 0u));   // {start, cum0 .. cum6}
        const u32 cum6 = rc[7], cum5 = rc[6], start = rc[0] + (TAIL && sgi ? T.tail_off : 0u);
        // sub-cells reached by any gated descriptor: bits 0..2 the thirds of the lower half, 3..5 of the upper
        u32 sub = 0, meta = (u32)c;
#pragma unroll
        for (int k = 0; k < KM; k++) {
          const bool lv = (gate[k] >> c) & 1u;     // (gate[k] = 0 for k >= K)
          const u32 zk = (reach[k] >> (8 + 3 * oz)) & 7u, yk = (reach[k] >> (2 * oy)) & 3u;
          const u32 sk = ((yk & 1u) ? zk : 0u) | ((yk & 2u) ? zk << 3 : 0u);
          sub |= lv ? sk : 0u;
          visits[k] += lv ? cum6 : 0u;           // the reference's loop visits every entry of every gated cell
          meta |= lv ? 0u : 0x100u << k;
        }
        const bool live = (gate_any >> c) & 1u;
        sub |= (sub & (sub >> 2) & 0x9u) << 1;   // per half the thirds from the first to the last reached one
        if (whole) {
          put(start, live ? cum6 : 0u, meta);                   // the whole bucket
          return;
        }
        if (!coarse) {
          // half h: entries before its first reached third (cum[i - 1], 0 for i = 0) and up to its last
          const u32 m0 = sub & 7u, m1 = sub & 0x38u;
          const u32 lo0 = (u32)__builtin_ctz(m0 | 0x40u), hi0 = 31u - (u32)__builtin_clz(m0 | 1u);
          const u32 lo1 = (u32)__builtin_ctz(m1 | 0x40u), hi1 = 31u - (u32)__builtin_clz(m1 | 1u);
          const u32 before0 = lo0 ? rc[lo0] : 0u, upto0 = rc[hi0 + 1];
          const u32 before1 = rc[lo1], upto1 = rc[hi1 + 1];
          put(start + before0, m0 ? upto0 - before0 : 0u, meta);
          put(start + before1, m1 ? upto1 - before1 : 0u, meta);
        } else {
          put(start, live ? cum5 : 0u, meta);                   // both halves
        }
        put(start + cum5, live ? cum6 - cum5 : 0u, meta);         // the overflow slice follows both halves
      };
#if SGTD_PLAN_SPARSE
      // every lane walks ITS cells — gated by one of its descriptors and holding a bucket in one of the segments, 7 of
      // the 27 on the synthetic maps — in ascending order; the wave runs as many trips as its busiest lane has cells
      u32 todo = gate_any & (ex_g[0] | ex_g[SEGS - 1]);
#if SGTD_EXP_PLAN_STAGE == 6
      todo = todo == 0x12345u ? 1u : 0u;
#endif
#pragma unroll 1
      while (todo) {
        const int c = __builtin_ctz(todo);
        todo &= todo - 1u;
#pragma unroll
        for (u32 sgi = 0; sgi < SEGS; sgi++) cell(c, sgi, (u32)__builtin_popcount(ex_g[sgi] & ((1u << c) - 1u)));
      }
#else
#pragma unroll 1
      for (int cs = 0; cs < (int)(SGTD_NCELL * SEGS); cs++) {
        const int c = TAIL ? cs >> 1 : cs;
        const u32 sgi = TAIL ? (u32)(cs & 1) : 0u;
        cell(c, sgi, rk[sgi]);
        rk[sgi] += (ex_g[sgi] >> c) & 1u;
      }
#endif
      const bool rec = emit && n != 0u;
      if (rec) {
        my_off = off;
        *wp = RangeWords{total, T.n_entries - total, SGTD_META_SENTINEL};   // the sentinel range (the main segment's sentinels)
        u32 hw[SGTD_PASS_HDR_WORDS];
#pragma unroll
        for (int i = 0; i < SGTD_PASS_HDR_WORDS; i++) hw[i] = 0;
        hw[0] = n | ((u32)(K == 3 ? 4 : K) << 8) | ((u32)K << 12);     // (three descriptors: swept with a fourth column that fails every gate)
        hw[1] = total;
#pragma unroll
        for (int k = 0; k < KM; k++) {
          hw[pass_hdr_word(PH_SLOT, k)] = d[k]; hw[pass_hdr_word(PH_FRAME, k)] = qfr[k];
          hw[pass_hdr_word(PH_Q0, k)] = hq[k][0]; hw[pass_hdr_word(PH_Q1, k)] = hq[k][1]; hw[pass_hdr_word(PH_Q2, k)] = hq[k][2];
          hw[pass_hdr_word(PH_LO2, k)] = hq[k][3]; hw[pass_hdr_word(PH_HI2, k)] = hq[k][4];
        }
        uint4 *h = P.pool + off;
#pragma unroll
        for (int i = 0; i < SGTD_PASS_HDR_UNITS; i++) h[i] = make_uint4(hw[4 * i], hw[4 * i + 1], hw[4 * i + 2], hw[4 * i + 3]);
      }
#pragma unroll
      for (int k = 0; k < KM; k++)
        if (k < K) {
          n_visit[d[k]] = visits[k];
          if (!rec) list[d[k]] = make_uint2(0u, 0u);
        }
    }
  }
  if (s < n_pass) P.rec_off[s] = my_off;
  }
}

// What the sweep holds about the pass it is working on
struct PassView {
  u32 hv;            // lane i: header word i (pass_hdr_word)
  u32 n, total;      // wave-uniform
  u32 k_real;        // descriptors of the pass (columns beyond have no list)
  // lane j: range j (lanes beyond the sentinel: offc = 0xFFFFFFFF)
  u32 offc, dlc, meta;
  __device__ __forceinline__ u32 word(int field, int k) const { return (u32)__builtin_amdgcn_readlane((int)hv, pass_hdr_word(field, k)); }
};

// vec with lane k replaced by the wave-uniform val
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int LANE>
__device__ __forceinline__ u32 write_lane(u32 vec, u32 val) {
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(val), "n"(LANE));     // (the lane as an inline constant: one scalar operand)
  return vec;
}

// a wave-uniform float computed by vector instructions, pinned to its vector register
__device__ __forceinline__ u32 in_vgpr_f(float x) {
  u32 v = __float_as_uint(x);
  asm volatile("" : "+v"(v));
  return v;
}

// a wave-uniform value held in a vector register (operand of v_cmp / v_pk_* without taking scalar registers)
__device__ __forceinline__ u32 in_vgpr(u32 x) {
  u32 v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
  return v;
}

// STDesc.cpp:372-399 for the K descriptor columns of one pass by one wavefront: streams the
// (union) visit list once, tests every entry against each descriptor, compacts each
// descriptor's matches in visit order into its own list.
// WIDE = false: the probe layout is below 4 GB, so entry addresses are a uniform base + a
// 32-bit byte offset (no quarter-rate 64-bit VALU address arithmetic per entry); the host
// picks the variant from the table size.  Record addresses are a wave-uniform list base + a
// 32-bit lane offset either way.
// FRAMES = false: no descriptor of the batch carries a frame id the table holds (the reference
// stamps every query descriptor with current_frame_id_, one beyond the map's last frame, quirk 1
// of SURVEY §8a), so the frame test of :373 is true for every entry and is not evaluated.
// Registers: the loop's scalar state is, per column, the list's base address and its match count;
// the thresholds and frames are wave-uniform values in VECTOR registers, the query sides scalar
// pairs (packed operands), the streams' cursors are parked in WaveSlab::state.
template <bool DIAG, bool WIDE, bool FRAMES, int K>
__device__ __forceinline__ void sweep_pass(const TableView &T, const ProbeBuffers &B, const QueryView &Q, double rough,
                                           const PassView &pv, u64 *bits, WaveSlab &slab, PendingLoads pending) {
  static_assert(!DIAG || K == 1, "the diagnostic sweep takes one descriptor at a time");
  constexpr int KP = (K + 1) / 2;      // packed pairs of descriptor columns
  const int lane = lane_id();
  const u32 id_bits = T.map.bits;
  const u32 total = pv.total;
  // the diagnostic sweep evaluates the reference's form verbatim on the exact f64 sides, :356-357
  double dq0 = 0.0, dq1 = 0.0, dq2 = 0.0, thr = 0.0;
  if constexpr (DIAG) {
    const QueryRec &r = Q.qrec[pv.word(PH_SLOT, 0)];
    dq0 = r.q0; dq1 = r.q1; dq2 = r.q2;
    thr = norm3(dq0, dq1, dq2) * rough;
  }
  // per lane: the gate penalties of its range (0 / +inf per descriptor) and its cell
  // ... and, in the product sweep, MINUS the column's upper squared threshold: the sums then run to d2 - hi2, "certainly
  // outside" is a compare with zero and "not certainly inside" a compare with -(hi2 - lo2) — one wave-uniform value for
  // all columns (the largest gap: an entry that is called undecided without being so is decided exactly like the others) —
  // so that no threshold stays in a register through the pass (eight vector registers less: six waves per SIMD)
  float penc[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    penc[k] = __uint_as_float((pv.meta >> (8 + k)) & 1u ? 0x7F800000u : 0u);
    if constexpr (!DIAG) penc[k] = penc[k] - __uint_as_float(pv.word(PH_HI2, k));      // (+inf stays +inf; hi2 is finite)
  }
  float ngap = 0.0f;      // -(largest hi2 - lo2 of the pass's columns), rounded away from zero
  if constexpr (!DIAG) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      const float g = __uint_as_float(pv.word(PH_HI2, k)) - __uint_as_float(pv.word(PH_LO2, k));
      ngap = !(g <= ngap) ? g : ngap;           // (max that keeps a NaN: then nothing is certainly inside)
    }
    ngap = -(ngap * 1.0001f + 1e-30f);
  }
  const u32 ngapv = in_vgpr_f(ngap);
  const u32 cellc = pv.meta & 0xFFu;
  // narrow layout: the address delta in bytes, so that an entry's byte offset is ONE three-operand add
  const u32 dlc_sel = WIDE ? pv.dlc : pv.dlc << 4;
  const u32 lane16 = (u32)lane << 4;
  // records of one descriptor are contiguous: make sure its stream's slab can take the worst
  // case (every visited entry matches)
  // A list is given the room its matches are EXPECTED to need (rec_rate / 256 of the visit list, measured
  // on the batch before, three times over) — not the worst case, every visit a match: with visit lists of
  // 10^4 entries and 8192 waves, worst-case reservations alone exceed the 32-bit record index.  A list that
  // does outgrow what its slab has left is moved to a fresh slab (relocate below, rare).
  bool fits = true;
  bool tight = false;   // some column's room is below the worst case: the groups check it
  u32 next0[K];         // granule of the first record of the column's list
#pragma unroll
  for (int k = 0; k < K; k++) next0[k] = 0;
  const u32 want = DIAG ? total : min(total, (total >> 8) * B.rec_rate + (((total & 255u) * B.rec_rate) >> 8) + 256u);
  auto granules = [](u32 records) { return (records >> SGTD_REC_SHIFT) + ((records & 3u) ? 1u : 0u); };
  auto room_of = [](u32 nxt, u32 end) { const u32 g = end - nxt; return g > (0xFFFFFFFFu >> SGTD_REC_SHIFT) ? 0xFFFFFFFFu : g << SGTD_REC_SHIFT; };   // records
  const u32 want_g = granules(want);
  // a slab of `take` GRANULES from the global cursor (0, 0: the buffer is exhausted)
  auto new_slab = [&](u32 take, u32 &nxt, u32 &end) {
    u64 got = 0;
    if (lane == 0) got = atomicAdd(B.rec_cursor(), (unsigned long long)take);
    got = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(got >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)got);
    if (got + take <= (u64)B.rec_cap) { nxt = (u32)got; end = (u32)got + take; }
    else { nxt = 0; end = 0; }
  };
  static_for<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if ((u32)k >= pv.k_real) return;      // (the fourth column of a pass of three has no list)
    u32 nxt = (u32)__builtin_amdgcn_readlane((int)slab.state, k), end = (u32)__builtin_amdgcn_readlane((int)slab.state, 4 + k);
    if (total && (u64)nxt + want_g > (u64)end) {
      // only the matches stay in a slab: slabs of 8 expected lists keep the space abandoned at a
      // slab's end to about an eighth however long the visit lists are
      new_slab(want_g > (1u << 28) ? want_g : max(B.rec_slab, 8u * want_g), nxt, end);
      slab.state = write_lane<k>(slab.state, nxt);
      slab.state = write_lane<4 + k>(slab.state, end);
    }
    fits = fits && ((u64)nxt + want_g <= (u64)end);
    tight = tight || (room_of(nxt, end) < total);
    slab.state = write_lane<8 + k>(slab.state, room_of(nxt, end));
    next0[k] = nxt;
  });
  if (!fits && lane == 0) B.overflow()[0] = 1;
  __builtin_amdgcn_wave_barrier();

  u32 matches[K];
  char *list_base[K];   // wave-uniform: the list's first record; record r of the list is at byte 4 r
#ifdef SGTD_EXP_SHADOW16
  // experiment (never shipped): a second, 2-byte stream of the records' frames behind the record buffer (the host
  // allocates half as much again): what the extra store per test costs the sweep
  char *shadow_base[K];
#endif
#pragma unroll
  for (int k = 0; k < K; k++) {
    matches[k] = 0;
    list_base[k] = reinterpret_cast<char *>(B.rec_at(next0[k]));
#ifdef SGTD_EXP_SHADOW16
    shadow_base[k] = reinterpret_cast<char *>(B.rec_at(B.rec_cap)) + 64 + ((size_t)next0[k] << (SGTD_REC_SHIFT + 1));
#endif
  }
  // wave-uniform constants of the columns in vector registers
  u32 qfv[K];
#pragma unroll
  for (int k = 0; k < K; k++) qfv[k] = (FRAMES || DIAG) ? in_vgpr(pv.word(PH_FRAME, k)) : 0u;
  const float nhi0 = -__uint_as_float(pv.word(PH_HI2, 0));      // (one column: the sum starts from it)
  f32x2 qx[KP], qy[KP], qz[KP];
  if constexpr (K >= 2) {
#pragma unroll
    for (int kp = 0; kp < KP; kp++) {
      qx[kp] = f32x2{__uint_as_float(pv.word(PH_Q0, 2 * kp)), __uint_as_float(pv.word(PH_Q0, 2 * kp + 1))};
      qy[kp] = f32x2{__uint_as_float(pv.word(PH_Q1, 2 * kp)), __uint_as_float(pv.word(PH_Q1, 2 * kp + 1))};
      qz[kp] = f32x2{__uint_as_float(pv.word(PH_Q2, 2 * kp)), __uint_as_float(pv.word(PH_Q2, 2 * kp + 1))};
    }
  }
  const float q0s = __uint_as_float(pv.word(PH_Q0, 0)), q1s = __uint_as_float(pv.word(PH_Q1, 0)), q2s = __uint_as_float(pv.word(PH_Q2, 0));
  const u32 n_words = (total + 63u) >> 6;
  slab.swept += total;
  // position -> range.  The starts of the non-empty ranges are marked in a bit array over the
  // positions of the visit list (this wave's 64 x 64-bit LDS window, rebuilt every 4096
  // positions; the mark of a range that starts at position p > 0 is bit p - 1): the range of
  // position 64 w + l is the number of ranges that start at or before 64 w (one compare against
  // the per-lane offsets + popcount, scalar) plus the marks of positions 64 w + 1 .. 64 w + l
  // (v_mbcnt over the window word) — one LDS read and one ds_bpermute per 64 entries (1 + K for
  // two or four columns: their gate penalties), whatever the number of ranges.
  const bool marks = (u32)lane <= pv.n && pv.offc != 0u;
  auto window = [&](u32 w_first) {     // w_first: a multiple of 64 words
    bits[lane] = 0;
    const u32 wr = ((pv.offc - 1u) >> 6) - w_first;
    if (marks && wr < 64u) atomicOr(reinterpret_cast<unsigned long long *>(bits + wr), 1ull << ((pv.offc - 1u) & 63u));
    __builtin_amdgcn_wave_barrier();
  };
  // one load group = NW consecutive words: located and their loads issued back to back (issue),
  // tested later (tests).  NW is a compile-time count: straight-line code.
  //   v      s0, s1, s2 (f32), id of the lane's entry per word
  //   pen    0 / +inf per column (the entry's cell passes the descriptor's gate or not), packed pairs
  //   cellv  cell (0..26) of the lane's entry (diagnostic sweep)
  auto issue = [&](auto &v, auto &pen, auto &cellv, u32 w0) {
    constexpr int NW = (int)(sizeof(v) / sizeof(v[0]));
    if ((w0 & 63u) == 0u) window(w0);           // (a group never straddles the window: w0 is a multiple of its size)
    PH_ADD(0);
    const u64 *bits_w = bits + (w0 & 63u);
#pragma unroll
    for (int u = 0; u < NW; u++) {
      const u32 w_lo = (w0 + u) << 6;
      const u64 bm = bits_w[u];                 // marks of positions w_lo + 1 .. w_lo + 64
      const u32 before = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pv.offc <= w_lo)) - 1u;
      const u32 j4 = __builtin_amdgcn_mbcnt_hi((u32)(bm >> 32), __builtin_amdgcn_mbcnt_lo((u32)bm, before)) << 2;
      const u32 dsel = (u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)dlc_sel);
      if (DIAG) cellv[u] = (u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)cellc);
      if constexpr (K >= 2) {
#pragma unroll
        for (int kp = 0; kp < KP; kp++) {
          pen[u][kp].x = __uint_as_float((u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)__float_as_uint(penc[2 * kp])));
          pen[u][kp].y = __uint_as_float((u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)__float_as_uint(penc[2 * kp + 1])));
        }
      }
      // entry = position + the range's delta; beyond the list: the sentinel entries
      const float4 *pa = WIDE ? reinterpret_cast<const float4 *>(T.ent + (w_lo + lane + dsel))
                              : reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(T.ent) + (dsel + lane16 + (w_lo << 4)));
#ifdef SGTD_EXP_L1
      pa = reinterpret_cast<const float4 *>(T.ent) + lane;     // experiment: every load hits the same L1-resident KB
#endif
      v[u] = *pa;
    }
    PH_ADD(1);
  };
  auto tests = [&](auto &v, auto &pen, auto &cellv, u32 w0) {
    constexpr int NW = (int)(sizeof(v) / sizeof(v[0]));
    u64 amb_any = 0;     // wave mask: some entry of the group fell between the two f32 thresholds
    // squared f32 distances of word u's entries to the K descriptors; the sums of two columns run
    // in the halves of packed f32 operations and start from the gate penalties
    auto dist2 = [&](int u, float (&d2)[K]) {
      if constexpr (K >= 2) {
        const f32x2 sx = {v[u].x, v[u].x}, sy = {v[u].y, v[u].y}, sz = {v[u].z, v[u].z};
#pragma unroll
        for (int kp = 0; kp < KP; kp++) {
          const f32x2 dx = qx[kp] - sx, dy = qy[kp] - sy, dz = qz[kp] - sz;
          f32x2 acc = __builtin_elementwise_fma(dx, dx, pen[u][kp]);      // pen 0: fl(dx * dx) as in f32_bounds
          acc = __builtin_elementwise_fma(dy, dy, acc);
          acc = __builtin_elementwise_fma(dz, dz, acc);
          d2[2 * kp] = acc.x; d2[2 * kp + 1] = acc.y;
        }
      } else {
        const float dx = q0s - v[u].x, dy = q1s - v[u].y, dz = q2s - v[u].z;
        d2[0] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, nhi0)));
      }
    };
    // one (word, column) test; PUSH = false: store the matches — `count` runs up; true: replay of
    // the group that only queues the provisional records (rare), words in reverse — `count` runs
    // back down to the record index of the word's first match
    auto test = [&](auto push_tag, int u, int k, float d2, u32 &count) mutable {
      constexpr bool PUSH = decltype(push_tag)::value;
      u32 id = __float_as_uint(v[u].w);
      // (the replay shares nothing with the first evaluation: no mask of the hot loop stays alive for it)
      if constexpr (PUSH) asm volatile("" : "+v"(id), "+v"(d2));
      // unsigned (src.frame_id_ - db.frame_id_) > 0  <=>  frame ids differ (:373)
      const bool other = !(FRAMES || DIAG) || qfv[k] != (id >> id_bits);
      bool hit, amb = false;
      u64 m;
      double dis = 0.0;
      if constexpr (DIAG) {   // the reference's form verbatim on the exact sides, :374-378
        hit = false;
        if ((w0 + u) * 64u + (u32)lane < total && other) {      // (not the sentinel entries: the exact test reads the cold table)
          const double *sp = T.cold_side + (size_t)id_entry(T.map, id) * 3;
          const double ex = dq0 - sp[0], ey = dq1 - sp[1], ez = dq2 - sp[2];
          dis = sqrt((ex * ex + ey * ey) + ez * ez);   // Eigen norm() association
          hit = dis < thr;
        }
        m = __builtin_amdgcn_ballot_w64(hit);
      } else {
        // (sentinel entries and gated-out cells: d2 = +inf, above every hi2 — f32_bounds keeps it finite)
        const bool near = !(d2 > 0.0f);                         // d2 is the squared distance MINUS hi2: not certainly outside (NaN stays in)
        hit = near && other;
        amb = hit && !(d2 < __uint_as_float(ngapv));            // not certainly inside either: provisional
        m = FRAMES ? __builtin_amdgcn_ballot_w64(near) & __builtin_amdgcn_ballot_w64(other)   // two plain compares: no mask round trip
                   : __builtin_amdgcn_ballot_w64(near);
      }
      if constexpr (PUSH) count -= (u32)__builtin_popcountll(m);
      // record index inside the list
      const u32 at = count + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
      if constexpr (!PUSH) {
        // wave-uniform base (the list) + a 32-bit lane offset: no 64-bit VALU address math
#ifdef SGTD_EXP_NOSTORE
        if (at == 0xFFFFFFF0u)
#endif
        if (hit && fits) *reinterpret_cast<u32 *>(list_base[k] + (at << 2)) = id;
#ifdef SGTD_EXP_SHADOW16
        if (hit && fits) *reinterpret_cast<unsigned short *>(shadow_base[k] + (at << 1)) = (unsigned short)(id >> id_bits);
#endif
        if constexpr (DIAG) {
          if (hit && fits) { B.rec_cell[B.rec_index(next0[k]) + at] = (unsigned char)cellv[u]; B.rec_dis[B.rec_index(next0[k]) + at] = dis; }
        }
        // amb_any |= m & ballot(!(d2 < lo2)) — as one unit, so that no hit mask outlives its test
        // (left to the scheduler, the masks of a whole group wait in scalar registers for this)
        if (!DIAG)
          asm volatile("v_cmp_ngt_f32 vcc, %1, %2\n\ts_and_b64 vcc, vcc, %3\n\ts_or_b64 %0, %0, vcc"
                       : "+s"(amb_any) : "v"(ngapv), "v"(d2), "s"(m) : "vcc");
        count += (u32)__builtin_popcountll(m);
      } else {
        if (amb && fits) {
          const u32 qa = atomicAdd(B.amb_count(), 1u);
          if (qa < B.amb_cap) B.amb_queue[qa] = make_uint2(at, pv.word(PH_SLOT, k));
          else B.overflow()[0] = 1;    // re-run with a larger queue (grows with the record buffer)
        }
      }
    };
    PH_ADD(2);
#pragma unroll
    for (int u = 0; u < NW; u++) {
      float d2[K];
      if (!DIAG) dist2(u, d2);
#pragma unroll
      for (int k = 0; k < K; k++) test(std::false_type{}, u, k, DIAG ? 0.0f : d2[k], matches[k]);
    }
#ifdef SGTD_EXP_NOSTORE
    amb_any = 0;      // (experiment build: nothing was stored, nothing to decide again)
#endif
    if (!DIAG && amb_any) {   // rare: about one in 10^4 matches
      u32 back[K];
#pragma unroll
      for (int k = 0; k < K; k++) back[k] = matches[k];
#pragma unroll
      for (int u = NW - 1; u >= 0; u--) {
        float d2[K];
        dist2(u, d2);
#pragma unroll
        for (int k = 0; k < K; k++) test(std::true_type{}, u, k, d2[k], back[k]);
      }
    }
    PH_ADD(3);
  };
  if (n_words) {
    // Groups of four words, window after window, then what is left in groups of 2 and 1 (inside the
    // last window).  ALL of a group's loads are waited for before its first record store: vmcnt
    // counts loads and stores in issue order, and the stores sit in branches the compiler cannot
    // count — left to itself it waits "all but the last 3 / 2 / 1 / 0 operations" before word
    // 0 / 1 / 2 / 3, which, once the stores of the words before are in the queue, means waiting for
    // their acknowledgements from L2.  (Measured and rejected: two-word groups double-buffered —
    // the wait for a group's loads then falls behind the stores of the group tested in between, whose
    // number the compiler cannot know either: 6.3 ms against 5.85; with unconditional stores into a
    // dump word, which it can count: 6.9 ms, three times the store instructions.  Eight-word groups
    // ahead of the four-word ones — fewer round trips per pass, 24 more vector registers: 5.76 ms
    // against 5.64.  The store as an asm unit that sets exec from the hit mask and back, without the
    // saved copy and the skip branch — four scalar instructions less per test: no difference, 5.07 ms
    // against 5.01.)
    // Room: a group adds at most 64 NW records to a list.  `safe` counts the four-word groups every list
    // still has room for (recomputed from the lists' real lengths when it runs out); a list without room
    // for the next group moves: a slab for what it holds + what the rest of the visit list is expected to
    // add, its records copied, the old ones abandoned.
    u32 safe = 0;
    auto make_room = [&](u32 w0) {
      safe = 0xFFFFFFFFu;
      static_for<K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if ((u32)k >= pv.k_real || !fits) return;
        u32 room = (u32)__builtin_amdgcn_readlane((int)slab.state, 8 + k);
        if (room - matches[k] < 256u) {
          const u32 rest = total - min(total, w0 << 6);          // visits still to come
          const u32 more = min(rest, (rest >> 8) * B.rec_rate + (((rest & 255u) * B.rec_rate) >> 8) + 512u);
          const u32 need = matches[k] + more, need_g = granules(need);
          u32 nxt, end;
          new_slab(max(B.rec_slab, need_g > (1u << 28) ? need_g : 4u * need_g), nxt, end);
          if (end - nxt < need_g) { fits = false; if (lane == 0) B.overflow()[0] = 1; return; }
          u32 *from = B.rec_at(next0[k]), *to = B.rec_at(nxt);
          __builtin_amdgcn_s_waitcnt(0x0F70);   // the list's records so far are in L2 ...
          for (u32 i = (u32)lane; i < matches[k]; i += SGTD_WAVE)
            to[i] = __hip_atomic_load(from + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... and are read from there
          next0[k] = nxt;
          list_base[k] = reinterpret_cast<char *>(to);
          if (lane == 0) atomicAdd(B.list_moves(), 1u);
          room = room_of(nxt, end);
          slab.state = write_lane<4 + k>(slab.state, end);
          slab.state = write_lane<8 + k>(slab.state, room);
        }
        safe = min(safe, (room - matches[k]) >> 8);
      });
      __builtin_amdgcn_s_waitcnt(0x0F70);   // (the copies have left before the list's next records are stored)
    };
    auto group = [&](auto nw_tag, u32 w0) {
      constexpr int NW = decltype(nw_tag)::value;
      float4 v[NW];
      f32x2 pen[NW][KP];
      u32 cellv[NW];
      issue(v, pen, cellv, w0);
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      tests(v, pen, cellv, w0);
    };
    u32 w0 = 0;
    bool touched = false;
    auto check_room = [&]() {   // (before every four-word group and once before the tail: at most 256 records per list either)
      if (tight) {
        if (safe == 0) make_room(w0);
        safe--;
      }
    };
    for (; w0 + 4u <= n_words; w0 += 4u) {
      check_room();
      group(std::integral_constant<int, 4>{}, w0);
      if (!touched) { pending.touch(); touched = true; }
    }
    const u32 left = n_words - w0;   // wave-uniform
    if (left) check_room();
    if (left & 2u) { group(std::integral_constant<int, 2>{}, w0); w0 += 2; }
    if (left & 1u) { group(std::integral_constant<int, 1>{}, w0); w0 += 1; }
    if (!touched) pending.touch();   // every path through the sweep leaves them complete
  } else {
    pending.touch();
  }
  // the pass's results: lane k stores for descriptor k
  {
    u32 r_slot = pv.hv, r_ptr = next0[0], r_match = matches[0];
    // (lane k's own header word is not the slot: fetch the K slot words by lane)
    r_slot = (u32)__builtin_amdgcn_ds_bpermute((pass_hdr_word(PH_SLOT, 0) + min(lane, K - 1)) << 2, (int)pv.hv);
#pragma unroll
    for (int k = 1; k < K; k++)
      if (lane == k) { r_ptr = next0[k]; r_match = matches[k]; }
#ifdef SGTD_EXP_NOSTORE
    r_match = 0;      // (experiment build: no record was stored — the lists stay empty for the kernels behind)
#endif
    if ((u32)lane < pv.k_real) B.list[r_slot] = make_uint2(r_ptr, fits ? r_match : 0u);
  }
  static_for<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if ((u32)k >= pv.k_real) return;
    if (!fits && lane == 0) atomicAdd(B.rec_need(), (unsigned long long)matches[k]);
    if (fits) slab.state = write_lane<k>(slab.state, next0[k] + granules(matches[k]));
  });
  __builtin_amdgcn_wave_barrier();
  PH_ADD(4);
}

#define SGTD_TICKET_MAX 64   // pass slots per ticket: one lane each in the ticket's offset load
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


#ifndef SGTD_SWEEP_WAVES
// at least five waves per SIMD (96 vector registers): the double-buffered groups hide the load latency
// of one wave, the other waves cover the LDS round trips of the locate step
#define SGTD_SWEEP_WAVES
#endif
template <bool DIAG, bool WIDE, bool FRAMES>
__global__ __launch_bounds__(SGTD_PROBE_THREADS) SGTD_SWEEP_WAVES void probe_sorted_kernel(
    TableView T, ProbeBuffers B, QueryView Q, PassPool P, double rough, const u32 *n_valid_p, const u32 *n_groups_p,
    u32 chunk /* 1..SGTD_TICKET_MAX */) {
  __shared__ u64 s_bits[SGTD_PROBE_THREADS / SGTD_WAVE][64];   // per wave: range starts of the current 4096 positions
  const int lane = lane_id();
  const u32 n_slots = pass_slot_count(*n_valid_p, *n_groups_p, !DIAG && SGTD_PAIR >= 2);
  // Every WAVE dequeues `chunk` consecutive pass slots at a time (no workgroup barrier).
  // Small chunks keep the passes in flight on one XCD — and with them the buckets it is
  // reading — within that XCD's 4 MB L2; the host sizes a ticket to about 1.5k entry visits.
  // Software pipeline per wave: the ticket after next is in flight, the next ticket's record
  // offsets are in flight, the next pass's header and ranges are in flight while the current
  // pass is swept.
  TicketQueue tq;
  tq.heads = B.xcd_heads();
  tq.n_chunks = (n_slots + chunk - 1) / chunk;
  u32 xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  tq.xcc = xcc & 7u;
  tq.select(0);
  WaveSlab slab{};
#ifdef SGTD_EXP_PHASE
  for (int i = 0; i < 8; i++) slab.ph[i] = 0;
  slab.t = __builtin_readcyclecounter();
  const u64 ph_start = slab.t;
#endif
  // lane i: record offset of pass slot i of ticket c
  auto load_offs = [&](u32 c) {
    u32 r = SGTD_NO_PASS;
    if (c != SGTD_NO_CHUNK && (u32)lane < chunk && c * chunk + (u32)lane < n_slots) r = P.rec_off[c * chunk + (u32)lane];
    return r;
  };
  u32 ov = load_offs(tq.resolve(tq.issue()));
  u32 tk_next = tq.issue();
  u32 nxt_c = tq.resolve(tk_next);
  tk_next = tq.issue();
  u32 ov_n = load_offs(nxt_c);
  u64 todo = __builtin_amdgcn_ballot_w64(ov != SGTD_NO_PASS);
  // the next pass of this wave's stream of tickets: its record offset, SGTD_NO_PASS at the end
  auto next_pass = [&]() -> u32 {
    while (!todo) {
      if (nxt_c == SGTD_NO_CHUNK) return SGTD_NO_PASS;
      ov = ov_n;
      nxt_c = tq.resolve(tk_next);      // requested one whole ticket ago
      tk_next = tq.issue();             // in flight during this ticket
      ov_n = load_offs(nxt_c);          // in flight during this ticket
      todo = __builtin_amdgcn_ballot_w64(ov != SGTD_NO_PASS);
    }
    const int i = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    return (u32)__builtin_amdgcn_readlane((int)ov, i);
  };
  // header (lane i < 16: word i) and ranges (lane j: range j) of the current and the next pass
  u32 hv = 0, r_off = 0xFFFFFFFFu, r_dl = 0, r_meta = SGTD_META_SENTINEL, hv_n = 0, rn_off = 0xFFFFFFFFu, rn_dl = 0, rn_meta = SGTD_META_SENTINEL;
  auto fetch = [&](u32 off, u32 &h, u32 &a, u32 &b, u32 &c) {
    if (off == SGTD_NO_PASS) return;
    h = reinterpret_cast<const u32 *>(P.pool + off)[lane & (SGTD_PASS_HDR_WORDS - 1)];
    // every lane loads 12 B whatever n is (no wait for the header): the lanes beyond the
    // sentinel read into the following records or the pool's slack and are masked below
    const RangeWords rw = reinterpret_cast<const RangeWords *>(P.pool + off + SGTD_PASS_HDR_UNITS)[lane];
    a = rw.offc; b = rw.dlc; c = rw.meta;
  };
  u32 off_c = next_pass();
  fetch(off_c, hv, r_off, r_dl, r_meta);
  while (off_c != SGTD_NO_PASS) {
    const u32 off_n = next_pass();
    fetch(off_n, hv_n, rn_off, rn_dl, rn_meta);          // in flight during this pass
    PendingLoads pend;
    pend.h = &hv_n; pend.a = &rn_off; pend.b = &rn_dl; pend.c = &rn_meta;
    PH_ADD(5);      // between passes: tickets, the next record's fetch
    const u32 w0 = (u32)__builtin_amdgcn_readlane((int)hv, 0);
    const u32 n = w0 & 0xFFu, kk = (w0 >> 8) & 0xFu;
    PassView pv;
    pv.hv = hv; pv.n = n; pv.total = (u32)__builtin_amdgcn_readlane((int)hv, 1); pv.k_real = w0 >> 12;
    {
      const bool mine = (u32)lane <= n;
      pv.offc = mine ? r_off : 0xFFFFFFFFu;
      pv.dlc = r_dl;
      pv.meta = mine ? r_meta : SGTD_META_SENTINEL;
    }
    auto pass = [&](auto k_tag) {
      constexpr int KK = decltype(k_tag)::value;
      sweep_pass<DIAG, WIDE, FRAMES, KK>(T, B, Q, rough, pv, s_bits[threadIdx.x >> 6], slab, pend);
    };
    if constexpr (!DIAG && SGTD_PAIR >= 4) { if (kk == 4) pass(std::integral_constant<int, 4>{}); }
    if constexpr (!DIAG && SGTD_PAIR >= 2) { if (kk == 2) pass(std::integral_constant<int, 2>{}); }
    if (kk == 1) pass(std::integral_constant<int, 1>{});
    off_c = off_n;
    hv = hv_n; r_off = rn_off; r_dl = rn_dl; r_meta = rn_meta;
  }
  if (lane == 0 && slab.swept) atomicAdd(B.swept(), (unsigned long long)slab.swept);
#ifdef SGTD_EXP_PHASE
  if (lane == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(&g_phase[i], slab.ph[i]);
    atomicAdd(&g_phase[6], 1ull);
    atomicAdd(&g_phase[7], __builtin_readcyclecounter() - ph_start);
  }
#endif
}

// The provisional records of the sweep, decided exactly (STDesc.cpp:374-378 in the squared,
// comparison-exact form): a record whose entry does not match after all gets the frame
// of no frame (SGTD_DEAD_ID: no vote, no candidate) and leaves the query's match count.
__global__ void resolve_undecided_kernel(TableView T, QueryView Q, ProbeBuffers B, u32 *q_M) {
  const u32 n = min(*B.amb_count(), B.amb_cap);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint2 it = B.amb_queue[i];
    const QueryRec &r = Q.qrec[it.y];
    u32 *rec = B.rec_at(B.list[it.y].x) + it.x;      // (index in the list: a list may have moved during its pass)
    const double *sp = T.cold_side + (size_t)id_entry(T.map, *rec) * 3;
    const double dx = r.q0 - sp[0], dy = r.q1 - sp[1], dz = r.q2 - sp[2];
    const double d2 = (dx * dx + dy * dy) + dz * dz;   // Eigen norm() association
    if (!(d2 < r.thr2)) {
      *rec = SGTD_DEAD_ID;
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
#define SGTD_TOPK_THREADS 1024     // (the two walks over a query's vote array are chains of loads: a thread of 1 024 walks a quarter of what one of 256 did)
__global__ __launch_bounds__(SGTD_TOPK_THREADS) void topk_kernel(const u32 *votes_all, u32 frame_span, u32 frame_lo,
                                                   int cand_num, int *n_cand, int *cand_frame,
                                                   int *cand_votes, unsigned char *slot_of_all) {
  __shared__ u64 red[SGTD_TOPK_THREADS / SGTD_WAVE];
  __shared__ int n_picked;
  __shared__ u32 s_hist[SGTD_TOPK_BINS];
  __shared__ u64 s_pool[SGTD_TOPK_POOL];
  __shared__ u32 s_thr, s_n_ge, s_npool;
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = lane_id();
  const u32 *votes = votes_all + (size_t)q * frame_span;
  unsigned char *slot_of = slot_of_all + (size_t)q * frame_span;
  for (int b = tid; b < SGTD_TOPK_BINS; b += SGTD_TOPK_THREADS) s_hist[b] = 0;
  if (tid == 0) { n_picked = 0; s_npool = 0; }
  __syncthreads();
  for (u32 f = tid; f < frame_span; f += SGTD_TOPK_THREADS) {
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
    for (u32 f = tid; f < frame_span; f += SGTD_TOPK_THREADS) {
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
        {   // the wave's largest key: the largest high word by one DPP reduction, then the largest low word among the lanes that hold
            // it (two reductions of six DPP steps: the twelve ds_bpermute of a 64-bit butterfly were most of a round's time)
          const u32 whi = wave_max_u32((u32)(best >> 32));
          const u32 wlo = wave_max_u32((u32)(best >> 32) == whi ? (u32)best : 0u);
          best = ((u64)whi << 32) | (u64)wlo;
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
    for (u32 f = threadIdx.x; f < frame_span; f += SGTD_TOPK_THREADS) {
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
      for (int w = 1; w < SGTD_TOPK_THREADS / SGTD_WAVE; w++) b = red[w] > b ? red[w] : b;
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

// The same sub-block as a stream of QUADS — four consecutive records of one list, the last quad
// of a list possibly short: one list search serves four records (the record buffer has room
// for the reads past a list's end).  s_pre: exclusive quad offsets, s_ptr: list starts, s_cnt:
// list lengths; returns the number of quads, `records` the number of records.
__device__ __forceinline__ u32 sub_open_quads(const QueryView &Q, const ProbeBuffers &B, int q, u32 d0, u32 cnt,
                                              u32 *s_pre /*[32]*/, u32 *s_ptr /*[32]*/, u32 *s_cnt /*[32]*/,
                                              u32 &visits, u32 &records) {
  const int lane = lane_id();
  u32 n = 0, p = 0, v = 0;
  if (lane < SGTD_SUB_DESCS && d0 + lane < cnt) {
    const long long d = (long long)q * Q.stride + d0 + lane;
    const uint2 lp = B.list[d];
    n = lp.y; p = lp.x; v = B.n_visit[d];
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

// quad r of the sub-block -> (descriptor index inside it, granule of its first record, records in it)
__device__ __forceinline__ void sub_locate_quad(const u32 *s_pre, const u32 *s_ptr, const u32 *s_cnt, u32 r, u32 &dd,
                                                u32 &addr, u32 &k) {
  u32 c = 0;
  if (s_pre[16] <= r) c = 16;
  if (s_pre[c + 8] <= r) c += 8;
  if (s_pre[c + 4] <= r) c += 4;
  if (s_pre[c + 2] <= r) c += 2;
  if (s_pre[c + 1] <= r) c += 1;
  dd = c;
  const u32 quad = r - s_pre[c];              // the quad inside its list
  addr = s_ptr[c] + quad;                     // a quad is a granule: the address stays one too
  k = min(4u, s_cnt[c] - (quad << 2));
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
  for (u32 d0 = d_first; d0 < min(d_first + Q.chunk, cnt); d0 += SGTD_SUB_DESCS) {
    u32 records;
    const u32 RQ = sub_open_quads(Q, B, q, d0, cnt, s_pre, s_ptr, s_cnt, visits, records);
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
#ifdef SGTD_EXP_VOTES16
        // experiment (never shipped): the pass as it would read a 2-byte frame stream — half the bytes per granule at
        // the same places; the halves of the records it finds there stand in for frames (the counts are wrong, the time is what is asked)
        const uint2 h = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(B.rec) + ((size_t)addr << 3));
        nrec[u] = make_uint4((h.x & 0xFFFFu) << B.id_bits, (h.x >> 16) << B.id_bits, (h.y & 0xFFFFu) << B.id_bits, (h.y >> 16) << B.id_bits);
#else
        nrec[u] = *reinterpret_cast<const uint4 *>(B.rec_at(addr));      // (a granule: 16-byte aligned)
#endif
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
  if (B.overflow()[0]) return;
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
  const u32 d_first = (u32)id.blk * Q.chunk;
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
#ifndef SGTD_VOTES_Q_THREADS
#define SGTD_VOTES_Q_THREADS 1024
#endif
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
  // workgroup b = query b / n_tiles, tile b % n_tiles: the tiles of a query are dispatched next to each other and read the
  // query's records at about the same time — from the Infinity Cache after the first of them (with the tile in blockIdx.y
  // a query's tiles ran a whole grid row apart: 100 000 frames, 768 queries: 7.1 ms for three full reads from HBM)
  const u32 n_tiles = (frame_span + tile_span - 1u) / tile_span;
  const int q = (int)(blockIdx.x / n_tiles);
  const u32 tile_id = blockIdx.x % n_tiles;
  const u32 tile_lo = tile_id * tile_span;
  const u32 n_bins = min(tile_span, frame_span - tile_lo);
  const bool dead = B.overflow()[0] != 0;     // the batch is re-run: leave zeros
  for (u32 f = tid; f < n_bins; f += SGTD_VOTES_Q_THREADS) s_hist[f] = 0;
  if (tid == 0) { s_M = 0; s_P = 0; }
  __syncthreads();
  const u32 cnt = Q.count[q];
  if (!dead) {
    u32 visits = 0, total = 0;
    for (int blk = wid; blk < blocks_per_query; blk += NW) {
      const u32 d_first = (u32)blk * Q.chunk;
      if (d_first >= cnt) break;
      votes_of_block<true>(Q, B, q, d_first, cnt, tile_lo, n_bins, s_hist, nullptr, s_pre[wid], s_ptr[wid], s_cnt[wid], visits, total);
    }
    if (lane == 0 && tile_id == 0) {
      atomicAdd(&s_M, total);
      atomicAdd(&s_P, (unsigned long long)visits);
    }
  }
  __syncthreads();
  u32 *votes = B.votes + (size_t)q * frame_span + tile_lo;
  for (u32 f = tid; f < n_bins; f += SGTD_VOTES_Q_THREADS) votes[f] = s_hist[f];
  if (tid == 0 && tile_id == 0) {      // resolve_undecided_kernel has already subtracted the records it killed
    atomicAdd(&q_M[q], s_M);
    atomicAdd(&q_P[q], s_P);
  }
}

// pass 1: blk_count[(q*blocks+blk)*64 + s] = matches of the block in slot s;
// also the per-query sums of visited entries / matches for the statistics
// compact list of one query's candidate matches, block after block in list order: the pairs
// (q_idx << 32 | g) of the records whose frame made the candidate list, and their slots.
// Written by block_count_kernel, consumed by block_write_kernel (no second record walk).
// NARROW (entry ids with at most 19 rank bits — frames of up to 524 288 entries): 32-bit words
// slot << 26 | descriptor index inside the block << 19 | rank of the entry among its frame's — the
// slot names the frame.
#define SGTD_NARROW_RANK_BITS 19
struct CompactLists {
  u64 *pair;             // [cap] slot << 58 | q_idx << 32 | entry id (NARROW: u32 words, see above)
  u32 *blk_start;        // [nq * blocks_per_query] first entry of the block's list
  u32 *blk_n;            // [nq * blocks_per_query] entries of the block's list
  u32 *cursor;           // global allocation cursor
  u32 cap;
};

// SLOT_TABLE: the frame -> candidate slot map of the query is the byte array topk_kernel wrote
// (slot_of, 0xFF = not a candidate), staged in LDS and indexed directly — one ds_read_u8 per
// record; for maps whose frame span does not fit LDS the 256-entry hash of the candidates
template <bool SLOT_TABLE, bool NARROW>
__global__ __launch_bounds__(256) void block_count_kernel(QueryView Q, ProbeBuffers B, const int *n_cand,
                                                          const int *cand_frame, int cand_num,
                                                          int blocks_per_query, u32 *blk_count, CompactLists L,
                                                          u32 *q_M, unsigned long long *q_P,
                                                          const unsigned char *slot_of_all, u32 frame_span, u32 frame_lo) {
  constexpr int NW = 256 / SGTD_WAVE;
  extern __shared__ unsigned char s_slot8[];   // [frame_span rounded up to 16] when SLOT_TABLE
  __shared__ u64 s_bits[NW][64];        // per wave: range starts of the current 4096 stream positions
  __shared__ u32 s_hist[NW][64];
  __shared__ u64 s_cand[SGTD_CAND_HASH];
  if (B.overflow()[0]) return;
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
  const u32 d_first = (u32)id.blk * Q.chunk;
  const size_t bslot = (size_t)q * blocks_per_query + id.blk;
  u32 *out = blk_count + bslot * 64;
  if (d_first >= cnt) { out[lane] = 0; if (lane == 0) { L.blk_start[bslot] = 0; L.blk_n[bslot] = 0; } return; }
  const u32 d_last = min(d_first + Q.chunk, cnt);
  // room for the block's compact list: at most every record of the block
  u32 nm = 0;
    for (u32 dd = d_first + lane; dd < d_last; dd += SGTD_WAVE) nm += B.list[(long long)q * Q.stride + dd].y;
  const u32 r_blk = wave_sum(nm);
  u32 start = 0;
  if (lane == 0) start = atomicAdd(L.cursor, r_blk);
  start = (u32)__builtin_amdgcn_readfirstlane((int)start);
  const bool fits = (unsigned long long)start + r_blk <= (unsigned long long)L.cap;
  if (!fits && lane == 0) B.overflow()[0] = 1;     // sized like the record buffer: grown and re-run with it
  s_hist[wid][lane] = 0;
  u32 visits = 0, total = 0, running = 0;
  // segment by segment: a candidate frame lives in one segment, so its matches still arrive in
  // (i, cell, j) order
  u64 *bits = s_bits[wid];
  for (u32 d0 = d_first; d0 < d_last; d0 += SGTD_SUB_DESCS) {
    // The records of the sub-block's 32 descriptors as ONE stream: the non-empty lists are its ranges
    // (range j in lane j: offset in the stream, address delta, descriptor), a record's range comes from
    // the marks of the range starts in an LDS bit window — the sweep's locate: one LDS read, two
    // v_mbcnt and two ds_bpermute per 64 records instead of a five-step binary search per record.
    u32 n = 0, p = 0, v = 0;
    if (lane < SGTD_SUB_DESCS && d0 + lane < cnt) {
      const long long d = (long long)q * Q.stride + d0 + lane;
      const uint2 lp = B.list[d];
      n = lp.y; p = lp.x; v = B.n_visit[d];
    }
    const u32 inc = wave_incl_scan(n);
    const u32 R = (u32)__builtin_amdgcn_readlane((int)inc, SGTD_WAVE - 1);
    visits += wave_sum(v);
    total += R;
    const u64 hm = __builtin_amdgcn_ballot_w64(n != 0u);
    const u32 nr = (u32)__builtin_popcountll(hm);
    const u32 jdst = (n != 0u ? __builtin_amdgcn_mbcnt_hi((u32)(hm >> 32), __builtin_amdgcn_mbcnt_lo((u32)hm, 0u)) : 63u) << 2;   // (lane 63 is nobody's range)
    const u32 pre = inc - n;
    const u32 offj = (u32)__builtin_amdgcn_ds_permute((int)jdst, (int)pre);
    const u32 grac = (u32)__builtin_amdgcn_ds_permute((int)jdst, (int)p);             // the range's list: its first granule (its stream offset: offj)
    const u32 ddc = (u32)__builtin_amdgcn_ds_permute((int)jdst, lane);
    const u32 offc = (u32)lane < nr ? offj : ((u32)lane == nr ? R : 0xFFFFFFFFu);
    const bool marks = (u32)lane <= nr && offc != 0u;
    auto window = [&](u32 w_first) {     // the marks of stream positions [64 w_first, 64 w_first + 4096)
      bits[lane] = 0;
      const u32 wr = ((offc - 1u) >> 6) - w_first;
      if (marks && wr < 64u) atomicOr(reinterpret_cast<unsigned long long *>(bits + wr), 1ull << ((offc - 1u) & 63u));
      __builtin_amdgcn_wave_barrier();
    };
    // the records of the next four words are loaded while the current four are looked up
    u32 nid[4], ndd[4];
    auto load4 = [&](u32 r0) {
      const u32 w0 = r0 >> 6;
      if ((w0 & 63u) == 0u) window(w0);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const u32 w_lo = (w0 + u) << 6;
        const u64 bm = bits[(w0 + u) & 63u];
        const u32 before = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(offc <= w_lo)) - 1u;
        const u32 j4 = __builtin_amdgcn_mbcnt_hi((u32)(bm >> 32), __builtin_amdgcn_mbcnt_lo((u32)bm, before)) << 2;
        const u32 gsel = (u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)grac), osel = (u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)offj);
        ndd[u] = (u32)__builtin_amdgcn_ds_bpermute((int)j4, (int)ddc);
        const u32 r = w_lo + lane;
        nid[u] = r < R ? B.rec_at(gsel)[r - osel] : B.rec[0];     // record = the list's first + the position inside the list
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
            if constexpr (NARROW)
              reinterpret_cast<u32 *>(L.pair)[pos] = (sl << 26) | ((d0 - d_first + dd[u]) << SGTD_NARROW_RANK_BITS) | (rid[u] & ((1u << B.id_bits) - 1u));
            else
              L.pair[pos] = ((u64)((sl << 26) | (d0 + dd[u])) << 32) | (u64)rid[u];
          }
        }
        running += (u32)__builtin_popcountll(m);
      }
    }
    __builtin_amdgcn_wave_barrier();     // (the next sub-block rebuilds the window)
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
__device__ __forceinline__ void batch_totals(const u32 *ctr, unsigned long long *tot) {
  tot[0] += 1ull;
  tot[1] += (ctr[10] | ctr[11]) ? 1ull : 0ull;
  tot[2] += (unsigned long long)ctr[9];
}
// (one query: its base in the pair buffer is 0 and the batch's total is its own — query_base_kernel's work, done here: one launch less)
__global__ __launch_bounds__(64) void block_scan_kernel(u32 *blk_count, int blocks_per_query, int cand_num,
                                                        const int *n_cand, long long *pair_off, u32 *q_pairs,
                                                        int *overflow, u32 *q_pair_base_of_one, u32 pair_cap,
                                                        unsigned long long *totals_of_one) {
  __shared__ u32 tot[64];
  const int q = blockIdx.x, s = threadIdx.x;
  if (overflow[0]) {
    if (totals_of_one && s == 0) batch_totals(reinterpret_cast<const u32 *>(overflow) - 10, totals_of_one);
    return;
  }
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
    if (q_pair_base_of_one) {
      q_pair_base_of_one[0] = 0u; q_pair_base_of_one[1] = acc;
      if (acc > pair_cap) overflow[1] = 1;
    }
    // (and the handle's running totals: every flag and count they read is final here — the list pass behind this kernel sets none)
    if (totals_of_one) batch_totals(reinterpret_cast<const u32 *>(overflow) - 10, totals_of_one);
  }
}

// behind a batch's last kernel: the handle's running totals (never reset: they also see the batches a caller enqueues
// one after the other without synchronising) — launches, launches that raised an overflow flag, match lists that moved
__global__ void batch_totals_kernel(const u32 *ctr, unsigned long long *tot) { batch_totals(ctr, tot); }

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
template <bool NARROW>
__global__ __launch_bounds__(256) void block_write_kernel(QueryView Q, ProbeBuffers B, CompactLists L,
                                                          int blocks_per_query,
                                                          const u32 *blk_excl, int cand_num,
                                                          const long long *pair_off, const u32 *q_pair_base,
                                                          u64 *pairs, IdMap map, const int *n_cand, const int *cand_frame) {
  constexpr int NW = 256 / SGTD_WAVE;
  constexpr int CAP = SGTD_WRITE_CAP;   // staged pairs per slot = one 128-B (16) or 64-B (8) line
  __shared__ u64 s_mask[NW][SGTD_WRITE_WORDS][64];   // per wave, word in flight and slot: lanes of the word that carry the slot
  __shared__ u64 s_stage[NW][64][CAP + 1];   // per wave and slot: pairs waiting for a full-line store (rows padded by one
                                             // word: a 128-byte row stride put every slot's k-th pair on the same banks)
  if (B.overflow()[0] || B.overflow()[1]) return;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const BlockId id = assemble_block(Q.n_queries, blocks_per_query);
  if (!id.valid) return;
  const int q = id.q;
  const size_t bslot = (size_t)q * blocks_per_query + id.blk;
  const u32 nv = L.blk_n[bslot];
  if (nv == 0) return;
  using Word = std::conditional_t<NARROW, u32, u64>;
  const Word *cp = reinterpret_cast<const Word *>(L.pair) + L.blk_start[bslot];
  const u32 d_first = (u32)id.blk * Q.chunk;
  // lane s carries, for candidate slot s, the output position of its first staged
  // pair (`running`) and the number of staged pairs (`fill`) — and, NARROW, the position of its
  // frame's first entry in the id map (a compact word names the frame by its slot)
  u32 running = 0, fill = 0, first_of_slot = 0;
  if (lane < cand_num)
    running = q_pair_base[q] + (u32)pair_off[(size_t)q * (cand_num + 1) + lane] + blk_excl[bslot * 64 + lane];
  if (NARROW && lane < cand_num && lane < n_cand[q]) first_of_slot = map.frame_first[(u32)cand_frame[(size_t)q * cand_num + lane] - map.frame_lo];
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
  Word nraw[SGTD_WRITE_WORDS];
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
      if constexpr (NARROW) {
        const u32 w = (u32)nraw[u];
        sl[u] = ok ? w >> 26 : 0xFFu;
        // q_idx << 32 | rank; the frame's first entry comes from the slot's lane
        pr[u] = ((u64)(d_first + ((w >> SGTD_NARROW_RANK_BITS) & 127u)) << 32) | (u64)(w & ((1u << SGTD_NARROW_RANK_BITS) - 1u));
        first[u] = (u32)__builtin_amdgcn_ds_bpermute((int)((sl[u] & 63u) << 2), (int)first_of_slot);
      } else {
        pr[u] = nraw[u] & 0x03FFFFFFFFFFFFFFull;
        sl[u] = ok ? (u32)(nraw[u] >> 58) : 0xFFu;
        first[u] = map.frame_first[(u32)nraw[u] >> map.bits];
      }
    }
    if (r0 + SGTD_WRITE_WORDS * SGTD_WAVE < nv) load2(r0 + SGTD_WRITE_WORDS * SGTD_WAVE);
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) {
      u32 g = first[u] + ((u32)pr[u] & ((1u << (NARROW ? SGTD_NARROW_RANK_BITS : map.bits)) - 1u));
      if (map.by_frame) g = map.by_frame[g];
      pr[u] = (pr[u] & 0xFFFFFFFF00000000ull) | (u64)g;
    }
    // lanes with equal slot, by commutative LDS ORs (the result does not depend on the order the hardware
    // applies them in) — for all words of the step at once: one LDS round trip, not one per word
    u64 gms[SGTD_WRITE_WORDS];
    u32 c_owns[SGTD_WRITE_WORDS];
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) s_mask[wid][u][lane] = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++)
      if (sl[u] != 0xFFu) atomicOr(&s_mask[wid][u][sl[u] & 63u], 1ull << lane);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) {
      gms[u] = sl[u] != 0xFFu ? s_mask[wid][u][sl[u] & 63u] : 0ull;
      c_owns[u] = (u32)__builtin_popcountll(s_mask[wid][u][lane]);     // pairs the word adds to slot == lane
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < SGTD_WRITE_WORDS; u++) {
      const bool valid = sl[u] != 0xFFu;
      const int s = (int)(sl[u] & 63u);
      const u64 gm = gms[u];
      const u32 c_own = c_owns[u];
      // rank = lanes below me in my group
      const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(gm >> 32), __builtin_amdgcn_mbcnt_lo((u32)gm, 0u)), count = (u32)__builtin_popcountll(gm);
#ifdef SGTD_EXP_WRITE_DIRECT
      // experiment: no staging lines — every pair goes straight to its list (runs of one slot are contiguous)
      const u32 base = __shfl(running, s);
      if (valid) pairs[base + rank] = pr[u];
      running += c_own;
      (void)count;
#else
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
#endif
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
    if (i < cnt) n = B.list[d].y;
    u32 tot;
    const u32 ex = block_excl_scan(n, lds, tot);
    if (i < cnt) {
      long long last = -1;    // key of the last record put out: cell << 32 | entry id
      for (u32 k = 0; k < n; k++) {
        long long best = 0x7FFFFFFFFFFFFFFFll;
        size_t at = 0;
        {
          const uint2 lp = B.list[d];
          const size_t p0 = B.rec_index(lp.x);
          const u32 m = lp.y;
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
