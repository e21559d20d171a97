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

// one table entry in the probe layout: 32 bytes in two 16-B halves kept in two arrays
// (head[E], tail[E]) so that each 16-B-per-lane load of a wavefront covers 1 KB of
// consecutive memory (8 full lines) instead of half of 16 lines
struct __attribute__((aligned(16))) HotHead {
  double s0, s1;      // side_length_ (scaled)
};
struct __attribute__((aligned(16))) HotTail {
  double s2;
  u32 frame;          // frame_id_
  u32 g;              // insertion index (bucket order == ascending g)
};
#define SGTD_HOT_BYTES 32   // per entry, both halves

// What the sweep needs about ONE query descriptor, 64 B: written per descriptor slot by the
// build kernel (gid, d unset), and per sorted position by sorted_desc_kernel.  A wavefront
// loads 16 of them with one 16-B-per-lane load (lane j = quarter j & 3 of record j >> 2):
//   quarter 0: q0, q1   quarter 1: q2, thr2   quarter 2: frame, gate mask, group id, slot d
struct __attribute__((aligned(64))) QueryRec {
  double q0, q1, q2;  // side_length_ (scaled)
  double thr2;        // exact squared match threshold (sq_threshold)
  u32 qframe;         // frame_id_
  u32 gate;           // 27-bit mask of the probe cells that pass the 1.5 gate (gate_mask)
  u32 gid;            // key-major: group (home cell) of the descriptor
  u32 d;              // key-major: descriptor slot
  u32 pad[4];
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

struct HashSlot {  // 16 bytes
  u64 key;
  u32 start;
  u32 len;
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

__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
  for (int d = SGTD_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
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
