// exchange_kernels.hip.h — the device side of the multi-GPU step: a rank's local
// top-candidate_num table leaves the query pipeline as soon as it is final (behind votes_topk_kernel /
// topk_kernel, before the match lists are written), travels in ONE all-gather per table group, and
// every rank merges the gathered tables with the reference's rule (src/sgtd/src/STDesc.cpp:423-433:
// the largest vote count first, ties -> the lowest frame id, a frame needs >= 5 votes) — the
// candidate list a single table over all shards gives, because the shards hold disjoint frame
// ranges and the global top-k of disjoint sets is the top-k of the union of the local top-k lists.
#pragma once
#include "common.hip.h"

// A rank's packed table as it travels: int32 frames[nq*cn] | votes[nq*cn] | flags[SGTD_XCHG_FLAG_WORDS]
//   flags[0] != 0  the batch outgrew a work buffer: its tables are not final (sgtd_sync re-runs it and
//                  exports again); a merge that saw such a table reports it in out_flags[0]
//   flags[1], [2]  nq, cn of the batch (a merge over tables of different shapes is refused)
//   flags[3]       serial number of the batch on its handle
#define SGTD_XCHG_FLAG_WORDS 4

__host__ __device__ __forceinline__ size_t xchg_packed_ints(int nq, int cn) { return (size_t)2 * nq * cn + SGTD_XCHG_FLAG_WORDS; }

__global__ __launch_bounds__(256) void export_candidates_kernel(const int *cand_frame, const int *cand_votes, const int *overflow,
                                                                int nq, int cn, u32 serial, int *packed) {
  const long long n = (long long)nq * cn;
  const bool dead = overflow[0] != 0;      // (the candidate kernels left zeros / garbage of the batch before)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    packed[i] = dead ? -1 : cand_frame[i];
    packed[n + i] = dead ? 0 : cand_votes[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int *f = packed + 2 * n;
    f[0] = dead ? 1 : 0; f[1] = nq; f[2] = cn; f[3] = (int)serial;
  }
}

// One wave per query: the n_tables * cn (votes, frame) keys of the query in registers (item i in lane i % 64),
// then the reference's arg-max rounds — the largest key, its holder cleared (match_array[...] = 0, :435) —
// until cn candidates are out or the best has fewer than min_votes votes.  Beside the merged list:
//   out_src[q][k]  = table << 8 | slot of merged candidate k in its owner's local table (-1: unused) — the owner holds
//                    the candidate's match list and its verification result
//   out_keep[q]    = bit s: slot s of table `my_table`'s local list is in the merged list (the rank's winners: lists /
//                    sgtd_verify_masked only run for them)
// Frames of different tables must be disjoint (they are: frame-range shards); equal keys collapse into one candidate.
#define SGTD_MERGE_PER_LANE 16      // n_tables * cn <= 1024

// PER: key registers per lane (the host picks the smallest of 1, 4, 8, 16 that holds n_tables * cn keys — a round costs
// PER compares to find the lane's best and PER to clear it: with 16 registers for the 50 keys of ONE table the kernel took
// 85 us for 2048 queries).  The largest 64-bit key of the wave: the largest high word (votes) by one DPP reduction, then
// the largest low word among its holders by another — two short chains instead of twelve ds_bpermute round trips.
template <int PER>
__global__ __launch_bounds__(256) void merge_candidates_kernel(const int *gathered, long long table_stride, int n_tables, int my_table,
                                                               int nq, int cn, int min_votes, int *out_frame, int *out_votes,
                                                               int *out_n, int *out_src, u64 *out_keep, int *out_flags) {
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const int q = blockIdx.x * (blockDim.x >> 6) + wid;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int bad = 0;
    for (int t = 0; t < n_tables; t++) {
      const int *f = gathered + (long long)t * table_stride + (long long)2 * nq * cn;
      if (f[0] != 0) bad |= 1;
      if (f[1] != nq || f[2] != cn) bad |= 2;
    }
    out_flags[0] = bad;
  }
  if (q >= nq) return;
  const int n_items = n_tables * cn;
  u64 a[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int i = j * SGTD_WAVE + lane;
    a[j] = 0;
    if (i < n_items) {
      const int t = i / cn, s = i - t * cn;
      const int *base = gathered + (long long)t * table_stride;
      const int f = base[(long long)q * cn + s];
      const int v = base[(long long)nq * cn + (long long)q * cn + s];
      if (f >= 0 && v >= min_votes && v > 0) a[j] = ((u64)(u32)v << 32) | (u64)(0xFFFFFFFFu - (u32)f);
    }
  }
  int picked = 0;
  u64 keep = 0;
  for (int round = 0; round < cn; round++) {
    u64 best = 0;
    int bj = 0;
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (a[j] > best) { best = a[j]; bj = j; }
    const u32 whi = wave_max_u32((u32)(best >> 32));
    if (whi == 0u) break;                                    // nothing with >= min_votes votes is left (:427,433)
    const u32 wlo = wave_max_u32((u32)(best >> 32) == whi ? (u32)best : 0u);
    const u64 wbest = ((u64)whi << 32) | (u64)wlo;
    const u64 holders = __ballot(best == wbest);
    const int L = __builtin_ctzll(holders);                  // (equal keys: the lowest item index names the source)
    const int item = __builtin_amdgcn_readlane(bj * SGTD_WAVE + lane, L);
#pragma unroll
    for (int j = 0; j < PER; j++) a[j] = (a[j] == wbest) ? 0ull : a[j];
    const int t = item / cn, s = item - t * cn;
    if (lane == 0) {
      out_frame[(long long)q * cn + picked] = (int)(0xFFFFFFFFu - wlo);
      out_votes[(long long)q * cn + picked] = (int)whi;
      out_src[(long long)q * cn + picked] = (t << 8) | s;
    }
    if (t == my_table) keep |= 1ull << s;
    picked++;
  }
  if (lane == 0) {
    out_n[q] = picked;
    out_keep[q] = keep;
    for (int k = picked; k < cn; k++) {
      out_frame[(long long)q * cn + k] = -1;
      out_votes[(long long)q * cn + k] = 0;
      out_src[(long long)q * cn + k] = -1;
    }
  }
}

// The verification results of the merged candidates from their owners' tables: gathered = per table
// [score f64 nq*cn | pose f64 nq*cn*12] (what sgtd_export_verify_dev wrote on every rank), src = the merge's
// out_src.  One thread per (query, merged candidate).
__global__ __launch_bounds__(256) void gather_verified_kernel(const double *gathered, long long table_stride, const int *src, int nq, int cn,
                                                              double *score, double *pose) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nq * cn) return;
  const int s = src[i];
  if (s < 0) {
    score[i] = -1.0;
    for (int k = 0; k < 12; k++) pose[i * 12 + k] = 0.0;
    return;
  }
  const long long q = i / cn;
  const double *base = gathered + (long long)(s >> 8) * table_stride;
  const long long j = q * cn + (s & 255);
  score[i] = base[j];
  const double *p = base + (long long)nq * cn + j * 12;
  for (int k = 0; k < 12; k++) pose[i * 12 + k] = p[k];
}

// Lists only for the rank's winners: the offsets of a query's match lists are the prefix sums of the votes of the
// local candidates that made the merged list (a candidate's votes ARE its list's length); the others get empty lists.
__global__ __launch_bounds__(256) void cand_prefix_masked_kernel(const int *n_cand, const int *cand_votes, const u64 *keep, int cand_num,
                                                                 int n_queries, long long *pair_off, u32 *q_pairs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_queries) return;
  const int nc = n_cand[q];
  const u64 m = keep[q];
  u32 acc = 0;
  for (int k = 0; k <= cand_num; k++) {
    pair_off[(size_t)q * (cand_num + 1) + k] = (long long)acc;
    if (k < nc && ((m >> k) & 1ull)) acc += (u32)cand_votes[(size_t)q * cand_num + k];
  }
  q_pairs[q] = acc;
}
