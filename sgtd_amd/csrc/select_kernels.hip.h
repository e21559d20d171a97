// select_kernels.hip.h — the passes over the match records behind the sweep
// (src/sgtd/src/STDesc.cpp:404-453) for batches with enough query frames to give every CU
// a query of its own: TWO launches, one workgroup per query each, instead of
// votes / topk / block_count / block_scan / block_write:
//
//   votes_topk_kernel   (:404-433) streams the query's match lists into an LDS vote histogram
//                       (the query's final match_array) and runs the reference's arg-max
//                       rounds right there.  A candidate's vote count IS the length of its
//                       match_list_ (every record votes once, :410-417, and belongs to the list
//                       of its frame, :437-449), so the offsets of the query's lists are the
//                       prefix sums of the candidates' votes — no counting pass.
//   pairs_query_kernel  (:434-449) streams the same lists once more, in list order, tile by
//                       tile of up to 8192 records: the tile's candidate records are
//                       compacted per wave in stream order, counted per (wave, slot), and
//                       scattered into ONE slot-sorted LDS image of the tile (a stable counting
//                       sort: equal-slot records keep the stream's order = the reference's
//                       (i, cell, j) order).  The image goes out with coalesced stores: the
//                       run of slot s continues that candidate's list where the tile before
//                       left it.  No compact intermediate list in HBM, no per-block offsets,
//                       and all partial output lines of a query are written by one CU within
//                       a few microseconds of each other (they merge in that XCD's L2).
//
// probe_kernels.hip.h keeps the five-kernel form for what these do not cover: small batches
// (one wave per 128-descriptor block fills the chip where one workgroup per query does not),
// frame spans beyond LDS, entry ids with more than 17 rank bits.
#pragma once
#include "probe_kernels.hip.h"

// ---------------------------------------------------------------------------
// votes + top-k of one query by ONE workgroup
// ---------------------------------------------------------------------------
#define SGTD_VT_THREADS 1024
#define SGTD_VT_BINS 4096      // vote-value histogram of the threshold search (values beyond: last bin)
// dynamic LDS: u32 hist[frame_span] | u32 vbins[SGTD_VT_BINS] | u64 pool[SGTD_TOPK_POOL]
__host__ __device__ __forceinline__ size_t votes_topk_lds_bytes(u32 frame_span) {
  return (((size_t)frame_span + 3) & ~(size_t)3) * 4 + SGTD_VT_BINS * 4 + SGTD_TOPK_POOL * 8;
}

__global__ __launch_bounds__(SGTD_VT_THREADS) void votes_topk_kernel(QueryView Q, ProbeBuffers B, u32 frame_span, u32 frame_lo,
                                                                      int blocks_per_query, int cand_num, u32 *q_M,
                                                                      unsigned long long *q_P, int *n_cand, int *cand_frame,
                                                                      int *cand_votes, long long *pair_off, u32 *q_pairs) {
  constexpr int NW = SGTD_VT_THREADS / SGTD_WAVE;
  extern __shared__ u32 s_dyn[];
  u32 *s_hist = s_dyn;
  u32 *s_vb = s_dyn + ((frame_span + 3u) & ~3u);
  u64 *s_pool = reinterpret_cast<u64 *>(s_vb + SGTD_VT_BINS);
  __shared__ u32 s_pre[NW][32];
  __shared__ u32 s_ptr[NW][32];
  __shared__ u32 s_cnt[NW][32];
  __shared__ u32 s_M, s_thr, s_n_ge, s_npool;
  __shared__ unsigned long long s_P;
  __shared__ u64 s_red[NW];
  __shared__ int s_picked;
  const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
  const int q = blockIdx.x;
  const bool dead = B.overflow()[0] != 0;     // the batch is re-run: leave zeros
  for (u32 f = tid; f < frame_span; f += SGTD_VT_THREADS) s_hist[f] = 0;
  for (u32 b = tid; b < SGTD_VT_BINS; b += SGTD_VT_THREADS) s_vb[b] = 0;
  if (tid == 0) { s_M = 0; s_P = 0; s_npool = 0; s_picked = 0; }
  __syncthreads();
  const u32 cnt = Q.count[q];
  if (!dead) {
    u32 visits = 0, total = 0;
    for (int blk = wid; blk < blocks_per_query; blk += NW) {
      const u32 d_first = (u32)blk * Q.chunk;
      if (d_first >= cnt) break;
      votes_of_block<true>(Q, B, q, d_first, cnt, 0u, frame_span, s_hist, nullptr, s_pre[wid], s_ptr[wid], s_cnt[wid], visits, total);
    }
    if (lane == 0) {
      atomicAdd(&s_M, total);
      atomicAdd(&s_P, (unsigned long long)visits);
    }
  }
  __syncthreads();
  // match_array of the query (sgtd_result_votes) and the vote-value histogram of the frames that can be candidates
  u32 *votes = B.votes + (size_t)q * frame_span;
  for (u32 f = tid; f < frame_span; f += SGTD_VT_THREADS) {
    const u32 v = s_hist[f];
    votes[f] = v;
    if (v >= 5) atomicAdd(&s_vb[min(v, (u32)SGTD_VT_BINS - 1u)], 1u);
  }
  if (tid == 0) {      // resolve_undecided_kernel has already subtracted the records it killed
    atomicAdd(&q_M[q], s_M);
    atomicAdd(&q_P[q], s_P);
  }
  __syncthreads();
  // ---- :423-433, as topk_kernel: largest t with count(votes >= t) >= cand_num (5 if fewer frames qualify at all)
  if (tid < SGTD_WAVE) {
    constexpr int PER = SGTD_VT_BINS / SGTD_WAVE;
    u32 mine = 0;
    for (int b = 0; b < PER; b++) mine += s_vb[lane * PER + b];
    u32 suffix = mine;   // inclusive suffix sum over lanes >= lane
#pragma unroll
    for (int d = 1; d < SGTD_WAVE; d <<= 1) {
      const u32 o = __shfl_down(suffix, d);
      if (lane + d < SGTD_WAVE) suffix += o;
    }
    const u64 okmask = __ballot(suffix >= (u32)cand_num);
    const u32 n_ge = __shfl(suffix, 0);
    if (okmask) {
      const int L = 63 - __builtin_clzll(okmask);
      if (lane == L) {
        u32 acc = suffix - mine;
        int b = L * PER + PER - 1;
        for (; b >= L * PER; b--) { acc += s_vb[b]; if (acc >= (u32)cand_num) break; }
        s_thr = (u32)max(b, 5); s_n_ge = acc;
      }
    } else if (lane == 0) { s_thr = 5; s_n_ge = n_ge; }
  }
  __syncthreads();
  const u32 thr = s_thr;
  int *cf = cand_frame + (size_t)q * cand_num, *cv = cand_votes + (size_t)q * cand_num;
  long long *po = pair_off + (size_t)q * (cand_num + 1);
  if (s_n_ge <= (u32)SGTD_TOPK_POOL) {
    for (u32 f = tid; f < frame_span; f += SGTD_VT_THREADS) {
      const u32 v = s_hist[f];
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
      u32 acc = 0;
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
          cf[picked] = (int)(f + frame_lo);
          cv[picked] = (int)(u32)(best >> 32);
          po[picked] = (long long)acc;
        }
        acc += (u32)(best >> 32);
        picked++;
      }
      if (lane == 0) {
        n_cand[q] = picked;
        q_pairs[q] = acc;
        for (int k = picked; k < cand_num; k++) { cf[k] = -1; cv[k] = 0; }
        for (int k = picked; k <= cand_num; k++) po[k] = (long long)acc;
      }
    }
    return;
  }
  // ---- general path: more ties at the threshold than the pool holds — the rounds run over the whole
  // histogram (a picked frame's count is zeroed like match_array's, :435)
  u32 acc = 0;
  for (int round = 0; round < cand_num; round++) {
    u64 best = 0;
    for (u32 f = tid; f < frame_span; f += SGTD_VT_THREADS) {
      const u64 key = ((u64)s_hist[f] << 32) | (u64)(0xFFFFFFFFu - f);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int dlt = SGTD_WAVE / 2; dlt > 0; dlt >>= 1) {
      const u64 o = __shfl_xor(best, dlt);
      best = o > best ? o : best;
    }
    if (lane == 0) s_red[wid] = best;
    __syncthreads();
    if (tid == 0) {
      u64 b = s_red[0];
      for (int w = 1; w < NW; w++) b = s_red[w] > b ? s_red[w] : b;
      const u32 v = (u32)(b >> 32);
      if (v >= 5) {
        const u32 f = 0xFFFFFFFFu - (u32)(b & 0xFFFFFFFFu);
        s_hist[f] = 0;
        cf[s_picked] = (int)(f + frame_lo);
        cv[s_picked] = (int)v;
        po[s_picked] = (long long)acc;
        acc += v;
        s_picked++;
      } else {
        s_picked |= 0x40000000;  // stop marker
      }
    }
    __syncthreads();
    if (s_picked & 0x40000000) break;
  }
  if (tid == 0) {
    const int picked = s_picked & 0x3FFFFFFF;
    n_cand[q] = picked;
    q_pairs[q] = acc;
    for (int k = picked; k < cand_num; k++) { cf[k] = -1; cv[k] = 0; }
    for (int k = picked; k <= cand_num; k++) po[k] = (long long)acc;
  }
}

// the lists' offsets from the candidates' votes where topk_kernel picked them (frame spans beyond LDS)
__global__ __launch_bounds__(256) void cand_prefix_kernel(const int *n_cand, const int *cand_votes, int cand_num, int n_queries,
                                                          long long *pair_off, u32 *q_pairs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_queries) return;
  const int nc = n_cand[q];
  u32 acc = 0;
  for (int k = 0; k <= cand_num; k++) {
    pair_off[(size_t)q * (cand_num + 1) + k] = (long long)acc;
    if (k < nc) acc += (u32)cand_votes[(size_t)q * cand_num + k];
  }
  q_pairs[q] = acc;
}

// ---------------------------------------------------------------------------
// match lists of one query by ONE workgroup (:434-449)
// ---------------------------------------------------------------------------
#ifndef SGTD_PQ_THREADS
#define SGTD_PQ_THREADS 512
#endif
#define SGTD_PQ_WAVES (SGTD_PQ_THREADS / SGTD_WAVE)
#ifndef SGTD_PQ_WORDS
#define SGTD_PQ_WORDS 4                                   // quad-words (64 lanes x 4 consecutive records of one list) per wave and tile
#endif
#ifndef SGTD_PQ_OCC
#define SGTD_PQ_OCC 4
#endif
#define SGTD_PQ_TILE_QUADS (SGTD_PQ_WAVES * SGTD_PQ_WORDS * SGTD_WAVE)
#define SGTD_PQ_WAVE_RECS (SGTD_PQ_WORDS * SGTD_WAVE * 4)  // records of a wave's share of a tile: its region of the dense image
#define SGTD_PQ_TILE_RECS (SGTD_PQ_WAVES * SGTD_PQ_WAVE_RECS)
#define SGTD_PQ_DESCS SGTD_PQ_THREADS                     // descriptors per super-block: one list per thread
#define SGTD_PQ_RANK_BITS (SGTD_PQ_DESCS <= 512 ? 17 : 16)  // image words: slot << 26 | descriptor in super-block << RANK_BITS | rank in frame
static_assert(SGTD_PQ_DESCS <= 1024 && SGTD_MAX_CAND <= 64, "an image word holds 6 slot bits and 26 - RANK_BITS descriptor bits");
static_assert(SGTD_PQ_WAVE_RECS * 4 >= 4 * 64 * 8 && (SGTD_PQ_WAVES & (SGTD_PQ_WAVES - 1)) == 0 && SGTD_PQ_WAVES >= 2,
              "a wave's region of the image holds the ranking masks of four dense words; image positions rotate over a power of two of regions");

// One non-empty list of the super-block as the tiles see it: its first quad in the super-block's stream of
// quads, its first record, its records, its descriptor (index inside the super-block).
struct __attribute__((aligned(16))) PqList { u32 pre, first, n, desc; };     // (pre: quads before the list, first: its first granule)

// dynamic LDS: u32 image[SGTD_PQ_TILE_RECS] | slot table u8[span rounded to 16, + 16] (SLOT_TABLE)
template <bool SLOT_TABLE>
__global__ __launch_bounds__(SGTD_PQ_THREADS) __attribute__((amdgpu_waves_per_eu(SGTD_PQ_OCC, SGTD_PQ_OCC))) void pairs_query_kernel(QueryView Q, ProbeBuffers B, const int *n_cand,
                                                                       const int *cand_frame, int cand_num,
                                                                       const long long *pair_off, const u32 *q_pair_base,
                                                                       u64 *pairs, IdMap map, u32 frame_span, u32 frame_lo,
                                                                       const u64 *keep) {
  constexpr int NW = SGTD_PQ_WAVES, QW = SGTD_PQ_WORDS;
  extern __shared__ __attribute__((aligned(16))) u32 s_img[];   // [SGTD_PQ_TILE_RECS]: per wave its dense candidates, then the tile's slot-sorted image
                                                                // (16-byte aligned: the ranking masks in it are 64-bit words)
  unsigned char *s_slot8 = reinterpret_cast<unsigned char *>(s_img + SGTD_PQ_TILE_RECS);
  __shared__ PqList s_ne[SGTD_PQ_DESCS];            // the super-block's non-empty lists, in order
  __shared__ u32 s_start[SGTD_PQ_DESCS + SGTD_WAVE + 8];   // their first quads, ascending (+ end markers: a wave reads the 64 starts behind its first list)
  __shared__ u32 s_cnt[NW][64];                     // candidates of the tile per (wave, slot)
  __shared__ u32 s_scan[NW + 1];
  __shared__ u64 s_cand[SGTD_CAND_HASH];
  if (B.overflow()[0] || B.overflow()[1]) return;
  const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
  const int q = blockIdx.x;
  const u32 cnt = Q.count[q];
  const int nc = n_cand[q];
  if (nc == 0 || cnt == 0) return;     // (workgroup-uniform)
  // lists only for the candidates in the mask (the multi-GPU step's winners, sgtd_finish_lists): the others get no
  // slot, their records pass like those of any other frame; pair_off holds the masked offsets
  const u64 kept = keep ? keep[q] : ~0ull;
  if (kept == 0) return;
  // frame -> candidate slot (the byte behind the span answers for the ids of dead records)
  if (SLOT_TABLE) {
    for (u32 f = tid; f < (((frame_span + 15u) & ~15u) + 16u) / 4u; f += SGTD_PQ_THREADS) reinterpret_cast<u32 *>(s_slot8)[f] = 0xFFFFFFFFu;
    __syncthreads();
    if (tid < nc && ((kept >> tid) & 1ull)) s_slot8[(u32)cand_frame[(size_t)q * cand_num + tid] - frame_lo] = (unsigned char)tid;
  } else {
    s_cand[tid & (SGTD_CAND_HASH - 1)] = SGTD_CAND_EMPTY;
    __syncthreads();
    if (tid < nc && ((kept >> tid) & 1ull)) {
      const u32 f = (u32)cand_frame[(size_t)q * cand_num + tid];
      u32 h = (f * 0x9E3779B1u) >> 24;
      while (atomicCAS(&s_cand[h], SGTD_CAND_EMPTY, ((u64)f << 8) | (u64)tid) != SGTD_CAND_EMPTY) h = (h + 1) & (SGTD_CAND_HASH - 1);
    }
  }
  // lane s of every wave: where candidate s's list continues (every wave keeps its own copy in step), the
  // insertion index of its frame's first entry (an image word names the frame by its slot)
  u32 out_next = 0, first_of_slot = 0;
  if (lane < nc) {
    out_next = q_pair_base[q] + (u32)pair_off[(size_t)q * (cand_num + 1) + lane];
    first_of_slot = map.frame_first[(u32)cand_frame[(size_t)q * cand_num + lane] - map.frame_lo];
  }
  const u32 id_bits = B.id_bits, rank_mask = (1u << id_bits) - 1u;
  const u64 lane_bit = 1ull << lane;
  u32 *my_img = s_img + wid * SGTD_PQ_WAVE_RECS;
  for (u32 sb0 = 0; sb0 < cnt; sb0 += SGTD_PQ_DESCS) {
    __syncthreads();      // (the tiles of the super-block before are done with the lists; the slot table is complete)
    // ---- the super-block's non-empty lists as one stream of quads
    u32 n = 0, p = 0;
    if (sb0 + tid < cnt) {
      const uint2 lp = B.list[(long long)q * Q.stride + sb0 + tid];
      p = lp.x; n = lp.y;
    }
    u32 RQ, K;
    const u32 pre = block_excl_scan((n + 3u) >> 2, s_scan, RQ);
    const u32 kx = block_excl_scan(n ? 1u : 0u, s_scan, K);
    if (n) { s_ne[kx] = PqList{pre, p, n, (u32)tid}; s_start[kx] = pre; }
    if ((u32)tid >= K) s_start[tid] = 0xFFFFFFFFu;
    if (tid < SGTD_WAVE + 8) s_start[SGTD_PQ_DESCS + tid] = 0xFFFFFFFFu;
    static_assert(SGTD_PQ_THREADS >= SGTD_WAVE + 8, "the end markers behind the list starts are written by the first 72 threads");
    __syncthreads();
    if (RQ == 0) continue;
    const u32 n_tiles = (RQ + SGTD_PQ_TILE_QUADS - 1) / SGTD_PQ_TILE_QUADS;
    // A wave's quads of tile t: [t * TILE_QUADS + wid * QW * 64, + QW * 64), word u = 64 consecutive quads.
    // Locate (the sweep's): k0 = the list that holds the wave's first quad (64 lanes x 8 starts compared at once),
    // lane l then holds the start of list k0 + 1 + l; the starts inside the wave's 256 quads become marks in one
    // 64-bit scalar per word (bit b of word u: a list starts at quad 64 u + b + 1), a quad's list is k0 + the
    // marks before it (scalar popcounts + v_mbcnt).
    uint4 nrec[QW];
    u32 nk[QW], ndd[QW];
    auto fetch = [&](u32 t) {
      const u32 r0 = t * SGTD_PQ_TILE_QUADS + (u32)wid * (QW * SGTD_WAVE);
      u32 below = 0;
#pragma unroll
      for (int i = 0; i < SGTD_PQ_DESCS / SGTD_WAVE; i++) below += s_start[i * SGTD_WAVE + lane] <= r0 ? 1u : 0u;     // (the starts ascend: their number is the index)
      const u32 k0 = wave_sum(below) - 1u;      // the first list starts at quad 0: at least one
      const u32 st = s_start[k0 + 1u + (u32)lane];
      u32 kl[QW];
      u32 c_prev = 0;
      bool slow = false;
#pragma unroll
      for (int u = 0; u < QW; u++) {
        const u32 w_lo = r0 + (u32)u * SGTD_WAVE;
        const u32 c = (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st <= w_lo + SGTD_WAVE));
        u64 marks = 0;
        for (u32 i = c_prev; i < c; i++) marks |= 1ull << ((u32)__builtin_amdgcn_readlane((int)st, (int)i) - 1u - w_lo);
        kl[u] = k0 + c_prev + __builtin_amdgcn_mbcnt_hi((u32)(marks >> 32), __builtin_amdgcn_mbcnt_lo((u32)marks, 0u));
        c_prev = c;
        slow = slow || c >= (u32)SGTD_WAVE;
      }
      if (slow) {   // more than 64 lists start inside the wave's quads (lists of a few records): a search per quad
#pragma unroll
        for (int u = 0; u < QW; u++) {
          const u32 r = r0 + (u32)u * SGTD_WAVE + (u32)lane;
          u32 lo = 0, hi = K;
          while (hi - lo > 1u) {
            const u32 mid = (lo + hi) >> 1;
            if (s_start[mid] <= r) lo = mid; else hi = mid;
          }
          kl[u] = lo;
        }
      }
#pragma unroll
      for (int u = 0; u < QW; u++) {
        const u32 r = r0 + (u32)u * SGTD_WAVE + (u32)lane;
        const PqList L = s_ne[min(kl[u], K - 1u)];
        const bool ok = r < RQ;
        const u32 quad = r - L.pre;                  // the quad inside its list (a quad is a granule of the record buffer)
        nk[u] = ok ? min(4u, L.n - (quad << 2)) : 0u;
        ndd[u] = L.desc << SGTD_PQ_RANK_BITS;
        nrec[u] = *reinterpret_cast<const uint4 *>(B.rec_at(ok ? L.first + quad : 0u));      // 16-byte aligned; the buffer has room for the reads past a list's end
      }
    };
    fetch(0);
    for (u32 t = 0; t < n_tiles; t++) {
      uint4 rc[QW];
      u32 kk[QW], dd[QW];
#pragma unroll
      for (int u = 0; u < QW; u++) { rc[u] = nrec[u]; kk[u] = nk[u]; dd[u] = ndd[u]; }
      if (t + 1 < n_tiles) fetch(t + 1);     // in flight while this tile is sorted
#if defined(SGTD_EXP_PQ) && SGTD_EXP_PQ == 2
      {   // experiment build (never shipped): the locate and the loads alone — 1.31 ms of the kernel's 2.68
        u32 x = 0;
        for (int u = 0; u < QW; u++) x ^= rc[u].x ^ rc[u].y ^ rc[u].z ^ rc[u].w ^ kk[u] ^ dd[u];
        if (x == 0x12345678u) pairs[0] = x;
        continue;
      }
#endif
      // ---- the wave's candidate records, in stream order (word, lane, record of the quad), as image words
      // in its own region of the image
      u32 nd = 0;      // wave-uniform: dense words so far
      static_assert(QW % 2 == 0, "the candidates of two quad-words are counted by one wave scan (16-bit halves)");
#pragma unroll
      for (int u0 = 0; u0 < QW; u0 += 2) {
        u32 sl[2][4], mine[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int u = u0 + h;
          const u32 w4[4] = {rc[u].x, rc[u].y, rc[u].z, rc[u].w};
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32 lf = w4[i] >> id_bits;                 // local frame (a dead record's is beyond every span)
            if (SLOT_TABLE) {
              sl[h][i] = (u32)s_slot8[min(lf, frame_span)];
              sl[h][i] = (u32)i < kk[u] ? sl[h][i] : 0xFFu;
            } else {
              sl[h][i] = 0xFFu;
              if ((u32)i < kk[u] && lf < frame_span) sl[h][i] = cand_slot(s_cand, lf + frame_lo);
            }
            mine[h] += sl[h][i] != 0xFFu ? 1u : 0u;
          }
        }
        // (a wave's candidates of one quad-word are at most 256: the two counts travel in the halves of one word)
        const u32 both = mine[0] | (mine[1] << 16);
        const u32 inc = wave_incl_scan(both);
        const u32 tot2 = (u32)__builtin_amdgcn_readlane((int)inc, SGTD_WAVE - 1);
        u32 at[2] = {nd + ((inc - both) & 0xFFFFu), nd + (tot2 & 0xFFFFu) + ((inc - both) >> 16)};
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int u = u0 + h;
          const u32 w4[4] = {rc[u].x, rc[u].y, rc[u].z, rc[u].w};
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (sl[h][i] != 0xFFu) my_img[at[h]++] = (sl[h][i] << 26) | dd[u] | (w4[i] & rank_mask);
        }
        nd += (tot2 & 0xFFFFu) + (tot2 >> 16);
      }
#if defined(SGTD_EXP_PQ) && SGTD_EXP_PQ == 3
      if (nd == 0xFFFFFFFFu) pairs[0] = 0;     // experiment build (never shipped): locate, loads and the candidates' dense words alone
      continue;
#endif
      __builtin_amdgcn_wave_barrier();
      // ---- dense words back into registers; the wave's region of the image then serves as scratch for the
      // ranking: per batch of eight dense words and slot the lanes of the word that carry the slot (commutative
      // LDS ORs: the result does not depend on the order the hardware applies them in).  A record's rank among the
      // wave's records of its slot = the slot's records in the words before (lane s counts slot s) + the lanes
      // before it in its own word's mask: stable in stream order.
      constexpr int MAXD = SGTD_PQ_WAVE_RECS / SGTD_WAVE, GRP = 4;     // (one branch per group of four dense words: a wave has 3.5 on average)
      u32 dw[MAXD], rk[MAXD];
      const u32 ndw = (nd + SGTD_WAVE - 1) / SGTD_WAVE;
#pragma unroll
      for (int g0 = 0; g0 < MAXD; g0 += GRP) {
#pragma unroll
        for (int w = g0; w < g0 + GRP; w++) { dw[w] = 0xFFFFFFFFu; rk[w] = 0; }
        if ((u32)g0 < ndw) {
#pragma unroll
          for (int w = g0; w < g0 + GRP; w++) {
            const u32 v = my_img[w * SGTD_WAVE + lane];          // (inside the wave's region whatever nd is)
            dw[w] = (u32)(w * SGTD_WAVE + lane) < nd ? v : 0xFFFFFFFFu;
          }
        }
      }
      u32 run_s = 0;        // lane s: the wave's records of slot s so far
      u64 *m = reinterpret_cast<u64 *>(my_img);      // [GRP][64]
#pragma unroll
      for (int g0 = 0; g0 < MAXD; g0 += GRP) {
        if ((u32)g0 < ndw) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int w = 0; w < GRP; w++) m[w * SGTD_WAVE + lane] = 0;
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int w = 0; w < GRP; w++)
            if (dw[g0 + w] != 0xFFFFFFFFu) atomicOr(&m[w * SGTD_WAVE + (dw[g0 + w] >> 26)], lane_bit);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int w = 0; w < GRP; w++) {
            const u64 own = m[w * SGTD_WAVE + lane];
            const u32 s = (dw[g0 + w] >> 26) & 63u;
            const u64 gm = m[w * SGTD_WAVE + s];
            const u32 before = (u32)__builtin_amdgcn_ds_bpermute((int)(s << 2), (int)run_s);
            rk[g0 + w] = before + __builtin_amdgcn_mbcnt_hi((u32)(gm >> 32), __builtin_amdgcn_mbcnt_lo((u32)gm, 0u));
            run_s += (u32)__builtin_popcountll(own);
          }
        }
      }
      s_cnt[wid][lane] = run_s;
      __syncthreads();
#if defined(SGTD_EXP_PQ) && SGTD_EXP_PQ == 4
      if (rk[0] == 0xFFFFFFF1u) pairs[0] = 0;     // experiment build (never shipped): everything up to the ranking and the first barrier
      continue;
#endif
      // ---- where the wave's records of slot s go in the image: behind the slots before s and the waves before it
      u32 tot = 0, mine_before = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const u32 c = s_cnt[w][lane];
        mine_before += w < wid ? c : 0u;
        tot += c;
      }
      const u32 slot_inc = wave_incl_scan(tot);
      const u32 slot_off = slot_inc - tot;                                   // lane s: first image position of slot s
      const u32 n_img = (u32)__builtin_amdgcn_readlane((int)slot_inc, SGTD_WAVE - 1);
      const u32 my_base = slot_off + mine_before;
      // image position p lives in the region of wave (p / 64) % NW, at word 64 (p / (64 NW)) + p % 64: every wave
      // flushes the 64-record chunks of its own region — nobody else reads them, so the next tile's dense words
      // can follow without a barrier
#pragma unroll
      for (int g0 = 0; g0 < MAXD; g0 += GRP) {
        if ((u32)g0 < ndw) {
#pragma unroll
          for (int w = g0; w < g0 + GRP; w++) {
            const u32 pos = (u32)__builtin_amdgcn_ds_bpermute((int)((dw[w] >> 26) << 2), (int)my_base) + rk[w];
            if (dw[w] != 0xFFFFFFFFu)
              s_img[((pos >> 6) & (NW - 1)) * SGTD_PQ_WAVE_RECS + (((pos >> 6) / NW) << 6) + (pos & 63u)] = dw[w];
          }
        }
      }
      __syncthreads();
#if defined(SGTD_EXP_PQ) && SGTD_EXP_PQ == 5
      out_next += tot;      // experiment build (never shipped): everything but the flush
      continue;
#endif
      // ---- the image goes out: position e of slot s's run continues candidate s's list
      for (u32 c = (u32)wid; c * SGTD_WAVE < n_img; c += NW) {      // (wave-uniform: the permutes read lanes 0..63)
        const u32 e = c * SGTD_WAVE + lane;
        const bool ok = e < n_img;
        const u32 wv = my_img[((c / NW) << 6) + lane];
        const u32 s4 = ok ? (wv >> 26) << 2 : 0u;
        const u32 o = (u32)__builtin_amdgcn_ds_bpermute((int)s4, (int)(out_next - slot_off));
        u32 g = (u32)__builtin_amdgcn_ds_bpermute((int)s4, (int)first_of_slot) + (wv & ((1u << SGTD_PQ_RANK_BITS) - 1u));
        if (ok) {
          if (map.by_frame) g = map.by_frame[g];
#if defined(SGTD_EXP_PQ) && SGTD_EXP_PQ == 1
          if (g == 0xFFFFFFF3u)      // experiment build (never shipped): no stores — 2.47 ms of 2.68
#endif
          pairs[o + e] = ((u64)(sb0 + ((wv >> SGTD_PQ_RANK_BITS) & (SGTD_PQ_DESCS - 1u))) << 32) | (u64)g;
        }
      }
      out_next += tot;
      __builtin_amdgcn_wave_barrier();
    }
  }
}
