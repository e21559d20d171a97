// common.hip.h — shared device helpers for the gfx950 kernels.
// wave = 64 lanes everywhere (CDNA4); compiled with -ffp-contract=off so that
// every f32/f64 multiply-add below is two individually rounded operations,
// exactly like the reference binary (built -O3 without -march, no FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SGTD_WAVE 64
#define SGTD_MAX_K 16
#define SGTD_NCELL 27
#define SGTD_MAX_CAND 64

typedef unsigned long long u64;
typedef unsigned int u32;

// constants of one engine, passed by value to kernels
struct DevCfg {
  int K;            // descriptor_near_num
  int tpi;          // triplets per keypoint = C(K-1,2)
  int cand_num;
  double min_len, max_len, scale, rough;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (SGTD_WAVE - 1); }

__device__ __forceinline__ u64 lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// Eigen 3.3 Vector3d::norm() association: sqrt((v0^2 + v1^2) + v2^2), each op
// rounded (STDesc.cpp:357,369,374-376)
__device__ __forceinline__ double norm3(double x, double y, double z) {
  return sqrt((x * x + y * y) + z * z);
}

// 12-bit label code, Combinatorial_Binary_Encoding (STDesc.cpp:3-16)
__host__ __device__ __forceinline__ u32 label_code(int a, int b, int c) {
  return ((u32)(a & 15) << 8) | ((u32)(b & 15) << 4) | (u32)(c & 15);
}

// 60-bit table key: code | x | y | z (16 bits each cell coordinate); ascending
// key order == (code, x, y, z) lexicographic, z-neighbours are adjacent keys
__host__ __device__ __forceinline__ u64 pack_key(u32 code, u32 x, u32 y, u32 z) {
  return ((u64)code << 48) | ((u64)x << 32) | ((u64)y << 16) | (u64)z;
}

__host__ __device__ __forceinline__ u64 mix64(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

// bucket hash of a packed table key: two 32-bit multiplies and two xor-shifts
// (the keys differ in a few low bits of four 16-bit fields; this spreads them
// well enough for a half-full open-addressing table)
__host__ __device__ __forceinline__ u32 hash_key(u64 k) {
  u32 h = (u32)k * 0x9E3779B1u ^ (u32)(k >> 32) * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}

// One table entry in the probe layout: 16 bytes — the three sides rounded to f32 and the entry's
// id.  The sweep decides almost every entry on these (f32_bounds below gives squared thresholds
// that make the f32 test conservative on both sides); the few entries that fall between them are
// decided on the exact f64 sides of the cold table, so every decision equals the reference's.
//
// id = (frame - frame_lo) << bits | i, i = rank of the entry among the entries of its frame in
// insertion order (IdMap): ONE 32-bit word names the frame (votes, candidate lookup: a shift) and
// the entry (only the candidates' pairs are ever translated back to the insertion index), so a
// match record is 4 bytes and the sweep reads nothing but the 16-B entries.
struct __attribute__((aligned(16))) HotEntry {
  float s0, s1, s2;   // (float)side_length_ (scaled)
  u32 id;             // frame_id_ and entry (IdMap)
};
#define SGTD_HOT_BYTES 16   // per entry
// id -> frame, insertion index.  Frames whose entries are contiguous in insertion order (maps
// built frame by frame): by_frame == nullptr and frame_first[f] is the insertion index of the
// frame's first entry; otherwise by_frame lists the insertion indices grouped by frame (stable)
// and frame_first[f] is the frame's first position in that list.
struct IdMap {
  const u32 *frame_first;   // [frame span]
  const u32 *by_frame;      // [n_entries] or nullptr
  u32 bits;                 // of i
  u32 frame_lo;
};
#define SGTD_DEAD_ID 0xFFFFFFFFu     // a record that turned out not to match; the sentinel entries' id
__device__ __forceinline__ u32 id_local_frame(const IdMap &m, u32 id) { return id >> m.bits; }
__device__ __forceinline__ u32 id_entry(const IdMap &m, u32 id) {
  const u32 p = m.frame_first[id >> m.bits] + (id & ((1u << m.bits) - 1u));
  return m.by_frame ? m.by_frame[p] : p;
}
// behind the last entry of a probe layout: entries with sides +inf that match nothing; the lanes of
// a sweep's last 64-entry word beyond the visit list read them (no validity test per entry)
#define SGTD_SENTINELS 64

// Inside a bucket (one reference cell + label code) the entries are partitioned into
// SGTD_YSLICES x SGTD_ZSLICES sub-cells — halves of the second side's cell interval, each cut
// into thirds of the third side's — plus one overflow slice, each in insertion order; sub-cell
// s = yh * SGTD_ZSLICES + zt, the thirds of one half are adjacent.  A query descriptor only
// visits the sub-cells its threshold box reaches (|side1 - q1| <= thr and |side2 - q2| <= thr:
// per half one contiguous run of thirds) and the overflow slice.  All entries of one map frame
// in one bucket that could match the same query descriptor lie in ONE slice
// (table_kernels.hip.h, slice_assign_kernel), so the matches of any (query descriptor, cell,
// frame) still come out in insertion order — the order of the reference's bucket scan
// restricted to a frame, which is all that votes and per-candidate match lists depend on.
#define SGTD_YSLICES 2
#define SGTD_ZSLICES 3
#define SGTD_NSUB (SGTD_YSLICES * SGTD_ZSLICES)
#define SGTD_NSLICE (SGTD_NSUB + 1)        // + overflow slice (always visited)
// one bucket of the directory, 32 B: cum[k] = entries of the bucket in slices 0..k
// (cum[SGTD_NSUB] = all of them, the reference's bucket length)
struct __attribute__((aligned(32))) BucketDir {
  u32 start;
  u32 cum[SGTD_NSLICE];
};
static_assert(sizeof(BucketDir) == 32 && SGTD_YSLICES == 2 && SGTD_ZSLICES == 3,
              "the sweep unpacks {start, cum[0..2]} and {cum[3..6]} from the two 16-B halves of a row");
#define SGTD_NRANGE (2 * SGTD_NCELL)       // lanes of a plan round: (cell, half) — or (cell, overflow) on the odd ones
#define SGTD_GROUP_ROW_BYTES 1024          // 27 BucketDir rows (864 B), padded

// What the sweep needs about ONE query descriptor, 64 B: written per descriptor slot by the
// build kernel (gid, d unset), and per sorted position by sorted_desc_kernel.  A wavefront
// loads 16 of them with one 16-B-per-lane load (lane j = quarter j & 3 of record j >> 2):
//   quarter 0: q0, q1   quarter 1: q2, thr2   quarter 2: frame, gate mask, group id, slot d
//   quarter 3: lo2, hi2 (f32 squared thresholds of the conservative f32 test), t_up
struct __attribute__((aligned(64))) QueryRec {
  double q0, q1, q2;  // side_length_ (scaled)
  double thr2;        // exact squared match threshold (sq_threshold)
  u32 qframe;         // frame_id_
  u32 gate;           // 27-bit mask of the probe cells that pass the 1.5 gate (gate_mask)
  u32 gid;            // key-major: group (home cell) of the descriptor
  u32 d;              // key-major: descriptor slot
  float lo2, hi2;     // f32 d2 below lo2: certainly a match; above hi2: certainly not (f32_bounds)
  float t_up;         // f32 upper bound of the match threshold (slice pruning in the sweep's plan)
  u32 pad;
};

// smallest y with sqrt_rn(y) >= thr:  (sqrt_rn(d2) < thr)  <=>  (d2 < y), because the
// correctly rounded sqrt is monotone.  Lets the per-entry test of STDesc.cpp:374-378
// compare squared distances bit-exactly, without a per-entry sqrt.
__device__ __forceinline__ double sq_threshold(double thr) {
  if (!(thr > 0.0)) return 0.0;              // dis < thr is never true
  double y = thr * thr;
  if (!(y > 0.0)) return y;                  // underflow: sqrt(0) = 0 < thr, nothing below 0
  if (!(y < __builtin_inf())) return y;      // overflow: every finite d2 is below
  while (y > 0.0 && sqrt(y) >= thr) y = __longlong_as_double(__double_as_longlong(y) - 1);
  while (true) {
    const double up = __longlong_as_double(__double_as_longlong(y) + 1);
    if (!(sqrt(up) < thr)) return up;
    y = up;
  }
}

// 27-bit mask of the probe cells that pass the gate ||side - centre|| < 1.5
// (STDesc.cpp:366-369); bit c = voxel_round index (x outer .. z inner, :327-333).
// sqrt_rn(v) < 1.5 <=> v < 2.25 exactly (sqrt(2.25) = 1.5, sqrt(pred(2.25)) rounds below 1.5).
__device__ __forceinline__ u32 gate_mask(double q0, double q1, double q2) {
  double ex[3], ey[3], ez[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double dx = q0 - ((double)(int)(q0 + (double)(k - 1)) + 0.5);
    const double dy = q1 - ((double)(int)(q1 + (double)(k - 1)) + 0.5);
    const double dz = q2 - ((double)(int)(q2 + (double)(k - 1)) + 0.5);
    ex[k] = dx * dx; ey[k] = dy * dy; ez[k] = dz * dz;
  }
  u32 m = 0;
#pragma unroll
  for (int c = 0; c < 27; c++)
    if (((ex[c / 9] + ey[(c / 3) % 3]) + ez[c % 3]) < 2.25) m |= 1u << c;
  return m;
}

// Squared thresholds for the sweep's f32 test.  The sweep computes, in f32 with individually
// rounded (or fused) operations, d2f = (dx^2 + dy^2) + dz^2 from dxk = fl32(fl32(q_k) - fl32(s_k)).
// With u = 2^-24, every visited entry within 2.5 of the query per axis (it lies in one of the 27
// neighbouring cells), and |s_k| <= |q_k| + 2.5:
//   |dxk - (q_k - s_k)| <= u (|q_k| + |s_k|)(1 + u) + u |q_k - s_k| (1 + u) <= u (4 |q_k| + 16) =: a_k
// so | ||dxf|| - d | <= A = sqrt(a_0^2 + a_1^2 + a_2^2) for the true distance d, and
// sqrt(d2f) is within (1 +- 2u) of ||dxf|| (three rounded operations on non-negative terms).
// The reference's own f64 value of d differs from the true one by less than 1e-15 relative.
// With the margin m = 2 A + 16 u thr + 1e-12 (twice what the bound needs):
//   d2f < lo2 = rounddown_f32((thr - m)^2)  ==>  the reference's dis < thr
//   d2f > hi2 = roundup_f32((thr + m)^2)    ==>  the reference's dis >= thr
// anything else (including NaN) is decided exactly.
// The product sweep does not keep lo2 and hi2 in registers: it starts the sum from -hi2,
//   acc = fma(dz, dz, fma(dy, dy, fma(dx, dx, -hi2)))        (from +inf - hi2 = +inf where the cell fails the gate),
// and tests acc > 0 (certainly outside) and acc < -g, g >= hi2 - lo2 (certainly inside; one g for the up to four
// descriptors of a pass: a larger g only sends more entries to the exact test).  acc differs from d2* - hi2 (d2* = the
// exact sum of the three squared f32 differences) by three roundings of partial sums of magnitude <= max(hi2, d2*):
// at either threshold at most 3 u hi2, i.e. 1.5 u thr on the distance, where the three rounded additions of the form
// above cost 2 u thr — both inside the 16 u thr that m reserves beyond the bound it needs.
__device__ __forceinline__ void f32_bounds(double q0, double q1, double q2, double thr, float &lo2, float &hi2) {
  const double u = 5.9604644775390625e-08;   // 2^-24
  const double a0 = u * (4.0 * fabs(q0) + 16.0), a1 = u * (4.0 * fabs(q1) + 16.0), a2 = u * (4.0 * fabs(q2) + 16.0);
  const double A = sqrt((a0 * a0 + a1 * a1) + a2 * a2);
  const double m = 2.0 * A + 16.0 * u * thr + 1e-12;
  const double lo = thr - m, hi = thr + m;
  float l = 0.0f;
  if (lo > 0.0) {
    const double l2 = lo * lo * (1.0 - 1e-15);
    l = (float)l2;
    if ((double)l > l2) l = __uint_as_float(__float_as_uint(l) - 1u);   // towards zero (l > 0 here)
  }
  const double h2 = hi * hi * (1.0 + 1e-15);
  float h = (float)h2;
  if ((double)h < h2) h = __uint_as_float(__float_as_uint(h) + 1u);     // upwards (h >= 0)
  // hi2 stays finite: the sweep gives d2f = +inf to what must never match (sentinel entries behind
  // the table, cells that fail a descriptor's gate); real entries have finite d2f (their sides and
  // the query's are below 65 541), so a threshold beyond the f32 range sends them all to the exact test
  if (!(h < __builtin_inff())) h = 3.4028234663852886e38f;
  if (!(thr > 0.0) || !(h2 == h2)) { l = 0.0f; h = 0.0f; }             // dis < thr is never true / NaN: nothing is a certain match,
  lo2 = l;                                                              // every d2f > 0 a certain miss, d2f == 0 decided exactly
  hi2 = h;
}

// the sweep record of one query descriptor (dis_threshold of STDesc.cpp:356-357 in squared,
// comparison-exact form; cell gate :366-369; f32 thresholds)
__device__ __forceinline__ void write_query_rec(QueryRec *out, double s0, double s1, double s2, double rough, u32 frame) {
  const double thr = norm3(s0, s1, s2) * rough;
  float lo2, hi2;
  f32_bounds(s0, s1, s2, thr, lo2, hi2);
  double2 *qr = reinterpret_cast<double2 *>(out);
  qr[0] = make_double2(s0, s1);
  qr[1] = make_double2(s2, sq_threshold(thr));
  reinterpret_cast<uint4 *>(qr)[2] = make_uint4(frame, gate_mask(s0, s1, s2), 0u, 0u);
  // upper bound of thr in f32 (relative margin, then rounded up): an entry whose third side is
  // further than this from q2 cannot match
  const double tm = thr * (1.0 + 1e-9) + 1e-12;
  float t_up = (float)tm;
  if ((double)t_up < tm) t_up = __uint_as_float(__float_as_uint(t_up) + 1u);
  if (!(tm == tm)) t_up = __builtin_inff();       // NaN threshold: prune nothing (nothing matches anyway)
  reinterpret_cast<uint4 *>(qr)[3] = make_uint4(__float_as_uint(lo2), __float_as_uint(hi2), __float_as_uint(t_up), 0u);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));   // operands of v_pk_*_f32

struct HashSlot {  // 16 bytes
  u64 key;
  u32 bucket;   // index into the bucket directory
  u32 len;      // entries of the bucket
};
#define SGTD_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

// inclusive wave scan (64 lanes), six DPP adds: row_shr 1/2/4/8 scan each 16-lane row,
// row_bcast:15 carries rows 0->1 and 2->3, row_bcast:31 carries the lower half into rows
// 2 and 3.  Lanes without a source read 0 (old = 0, bound_ctrl off).  All 64 lanes must
// be active.
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (u32)x;
}

// sum over the 64 lanes, the same value in every lane: the inclusive DPP scan's last lane holds
// it (six DPP adds and a v_readlane: no LDS crossbar round trips).  All 64 lanes must be active.
__device__ __forceinline__ u32 wave_sum(u32 v) {
  return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan(v), SGTD_WAVE - 1);
}

// largest value over the 64 lanes, in every lane (the DPP steps of wave_incl_scan with max for +; 0 is the identity)
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
  int x = (int)v;
#define SGTD_MAX_STEP(ctrl, rows) { const u32 o = (u32)__builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xf, false); x = (int)((u32)x > o ? (u32)x : o); }
  SGTD_MAX_STEP(0x111, 0xf) SGTD_MAX_STEP(0x112, 0xf) SGTD_MAX_STEP(0x114, 0xf) SGTD_MAX_STEP(0x118, 0xf)
  SGTD_MAX_STEP(0x142, 0xa) SGTD_MAX_STEP(0x143, 0xc)
#undef SGTD_MAX_STEP
  return (u32)__builtin_amdgcn_readlane(x, SGTD_WAVE - 1);
}

// Stable multi-split rank inside one wave: lanes with equal `digit` (BITS wide)
// among the `valid` lanes form a group; returns the lane's rank inside its
// group (in lane order) and the group size.  BITS ballots, no LDS.
template <int BITS>
__device__ __forceinline__ void wave_group_rank(u32 digit, bool valid, u32 &rank, u32 &count) {
  u64 m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; b++) {
    u64 bal = __ballot((digit >> b) & 1u);
    m &= ((digit >> b) & 1u) ? bal : ~bal;
  }
  rank = __popcll(m & lanemask_lt());
  count = __popcll(m);
}
