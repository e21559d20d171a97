// probe_kernels.hip.h — candidate_selector (src/sgtd/src/STDesc.cpp:318-460)
//
//   probe     (:351-400) one wavefront per query descriptor: 27 lanes resolve
//             the 27 cells (truncating (int)(side+inc), gate ||side-centre||<1.5,
//             hash lookup key -> bucket), then all 64 lanes stream the
//             concatenated bucket ranges coalesced from the 28-B/entry hot
//             arrays; __ballot of the match predicate is stored as one 64-bit
//             mask word per 64 visited entries; votes by atomicAdd (:404-420)
//   offsets   per-query exclusive scan of the match counts => deterministic
//             positions in the reference's (i, cell, j) order
//   emit      replays the mask words (no table reads for the predicate) and
//             writes the ordered rough-match records by ballot/popcount prefix
//   topk      (:423-433) candidate_num rounds of arg-max over the vote
//             histogram: votes desc, frame id asc, stop below 5 votes
//   assemble  (:434-449) stable split of the ordered records by candidate slot
//             (wave_group_rank) => match_list_ of every candidate in order
#pragma once
#include "common.hip.h"

struct TableView {
  const double *s0, *s1, *s2;  // [E] sorted order
  const u32 *frame;            // [E]
  const u32 *perm;             // [E] sorted position -> insertion index
  const HashSlot *hash;
  u32 hash_mask;
  u32 n_entries;
  u32 frame_lo;                // votes are indexed by frame - frame_lo
  u32 frame_span;              // number of vote bins per query
};

struct QueryView {
  const double *side;   // [n_slots*3]
  const int *label;     // [n_slots*3]
  const u32 *frame;     // [n_slots]
  const u32 *count;     // [n_queries] descriptors per query
  long long stride;     // descriptor slots per query
  int n_queries;
};

struct ProbeBuffers {
  u64 *mask_words;      // [mask_cap]
  u32 mask_cap;
  u32 *mask_cursor;     // global slab cursor
  u32 *mask_ptr;        // [n_slots] first mask word of descriptor
  u32 *n_visit;         // [n_slots] entries visited by descriptor (T_d)
  u32 *n_match;         // [n_slots] matches of descriptor
  u32 *votes;           // [n_queries * frame_span]
  int *overflow;        // [2]: 0 mask words, 1 records
};

#define SGTD_PROBE_THREADS 256
#define SGTD_MASK_SLAB 512u   // mask words a wave takes from the global cursor at once

// resolves the 27 cells of one query descriptor; lane c < 27 returns its
// bucket (start,len) (len = 0 if gated out / absent) — STDesc.cpp:358-371
__device__ __forceinline__ void resolve_cells(const TableView &T, double q0, double q1, double q2,
                                              u32 code, u32 &start, u32 &len) {
  const int c = lane_id();
  start = 0; len = 0;
  if (c < SGTD_NCELL) {
    const int ix = c / 9 - 1, iy = (c / 3) % 3 - 1, iz = c % 3 - 1;  // voxel_round order (:327-333)
    const int x = (int)(q0 + (double)ix), y = (int)(q1 + (double)iy), z = (int)(q2 + (double)iz);
    const double cx = (double)x + 0.5, cy = (double)y + 0.5, cz = (double)z + 0.5;
    const bool gate = norm3(q0 - cx, q1 - cy, q2 - cz) < 1.5;        // :366-369
    if (gate && x >= 0 && y >= 0 && z >= 0 && x < 65536 && y < 65536 && z < 65536) {
      const u64 key = pack_key(code, (u32)x, (u32)y, (u32)z);
      u32 h = (u32)mix64(key) & T.hash_mask;
      while (true) {
        const HashSlot s = T.hash[h];
        if (s.key == key) { start = s.start; len = s.len; break; }
        if (s.key == SGTD_EMPTY_KEY) break;
        h = (h + 1) & T.hash_mask;
      }
    }
  }
}

// position `pos` in the concatenation of the 27 ranges -> (cell, entry index)
__device__ __forceinline__ void locate(const u32 *cell_off /*[28] LDS*/, const u32 *cell_start,
                                       u32 pos, int &cell, u32 &entry) {
  // branch-free binary search for the last cell with off <= pos (offsets ascending)
  int c = 0;
  if (cell_off[c + 16] <= pos) c += 16;
  if (cell_off[c + 8] <= pos) c += 8;
  if (cell_off[c + 4] <= pos) c += 4;
  if (cell_off[c + 2] <= pos) c += 2;
  if (cell_off[c + 1] <= pos) c += 1;
  cell = c;
  entry = cell_start[c] + (pos - cell_off[c]);
}

__global__ __launch_bounds__(SGTD_PROBE_THREADS) void probe_kernel(TableView T, QueryView Q,
                                                                   ProbeBuffers B, double rough) {
  constexpr int NW = SGTD_PROBE_THREADS / SGTD_WAVE;
  __shared__ u32 s_off[NW][32];    // exclusive offsets, padded to 32 with UINT_MAX
  __shared__ u32 s_start[NW][32];
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const long long n_waves = (long long)gridDim.x * NW;
  const long long n_slots = (long long)Q.n_queries * Q.stride;
  u32 slab_next = 0, slab_end = 0;  // this wave's private mask-word slab

  for (long long d = (long long)blockIdx.x * NW + wid; d < n_slots; d += n_waves) {
    const int q = (int)(d / Q.stride);
    const u32 i = (u32)(d - (long long)q * Q.stride);
    if (i >= Q.count[q]) continue;   // wave-uniform
    const double q0 = Q.side[d * 3 + 0], q1 = Q.side[d * 3 + 1], q2 = Q.side[d * 3 + 2];
    const u32 code = label_code(Q.label[d * 3 + 0], Q.label[d * 3 + 1], Q.label[d * 3 + 2]);
    const u32 qframe = Q.frame[d];
    const double thr = norm3(q0, q1, q2) * rough;   // :356-357

    u32 start, len;
    resolve_cells(T, q0, q1, q2, code, start, len);
    const u32 inc = wave_incl_scan(len);
    const u32 total = __shfl(inc, SGTD_WAVE - 1);
    if (lane < 32) {
      s_off[wid][lane] = (lane < SGTD_NCELL) ? inc - len : 0xFFFFFFFFu;
      s_start[wid][lane] = start;
    }
    const u32 n_words = (total + 63u) >> 6;
    // mask words from the wave-private slab (one global atomic per SGTD_MASK_SLAB words)
    u32 mbase = 0;
    if (n_words) {
      if (slab_next + n_words > slab_end) {
        u32 take = n_words > SGTD_MASK_SLAB ? n_words : SGTD_MASK_SLAB;
        u32 got = 0;
        if (lane == 0) got = atomicAdd(B.mask_cursor, take);
        got = __shfl(got, 0);
        slab_next = got; slab_end = got + take;
      }
      mbase = slab_next;
      slab_next += n_words;
    }
    const bool fits = (unsigned long long)mbase + n_words <= (unsigned long long)B.mask_cap;
    if (!fits && lane == 0) B.overflow[0] = 1;
    __builtin_amdgcn_wave_barrier();

    u32 matches = 0;
    u32 *votes = B.votes + (size_t)q * T.frame_span;
    for (u32 w = 0; w < n_words; w++) {
      const u32 pos = (w << 6) + lane;
      bool hit = false;
      if (pos < total) {
        int cell; u32 e;
        locate(s_off[wid], s_start[wid], pos, cell, e);
        const double dx = q0 - T.s0[e], dy = q1 - T.s1[e], dz = q2 - T.s2[e];
        const u32 fr = T.frame[e];
        // unsigned (src.frame_id_ - db.frame_id_) > 0  <=>  ids differ (:373)
        hit = (qframe != fr) && (norm3(dx, dy, dz) < thr);        // :374-378
        if (hit) atomicAdd(&votes[fr - T.frame_lo], 1u);          // :410
      }
      const u64 m = __ballot(hit);
      matches += __popcll(m);
      if (lane == 0 && fits) B.mask_words[mbase + w] = m;
    }
    if (lane == 0) {
      B.mask_ptr[d] = mbase;
      B.n_visit[d] = total;
      B.n_match[d] = fits ? matches : 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// per query: exclusive scan of n_match over its descriptors -> rec_off[d]
// (relative to the query), totals M_q and P_q; one workgroup per query
__global__ __launch_bounds__(256) void query_offsets_kernel(QueryView Q, const u32 *n_match,
                                                            const u32 *n_visit, u32 *rec_off,
                                                            u32 *q_M, unsigned long long *q_P) {
  __shared__ u32 lds[256 / SGTD_WAVE + 1];
  const int q = blockIdx.x;
  const u32 cnt = Q.count[q];
  const size_t base = (size_t)q * Q.stride;
  u32 carry = 0;
  unsigned long long visits = 0;
  for (u32 i0 = 0; i0 < cnt; i0 += 256) {
    const u32 i = i0 + threadIdx.x;
    const u32 v = (i < cnt) ? n_match[base + i] : 0;
    if (i < cnt) visits += n_visit[base + i];
    u32 tot;
    const u32 ex = block_excl_scan(v, lds, tot);
    if (i < cnt) rec_off[base + i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) q_M[q] = carry;
  // block sum of visits
  __shared__ unsigned long long vs[256];
  vs[threadIdx.x] = visits;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) vs[threadIdx.x] += vs[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) q_P[q] = vs[0];
}

// exclusive scan of q_M over queries (n_queries small): q_base[q], q_base[n] = total
__global__ __launch_bounds__(256) void query_base_kernel(const u32 *q_M, u32 *q_base, int n_queries,
                                                         u32 rec_cap, int *overflow) {
  __shared__ u32 lds[256 / SGTD_WAVE + 1];
  u32 carry = 0;
  for (int q0 = 0; q0 < n_queries; q0 += 256) {
    const int q = q0 + threadIdx.x;
    const u32 v = (q < n_queries) ? q_M[q] : 0;
    u32 tot;
    const u32 ex = block_excl_scan(v, lds, tot);
    if (q < n_queries) q_base[q] = carry + ex;
    // u32 overflow of the running sum would corrupt offsets: flag it
    if (carry + tot < carry && threadIdx.x == 0) overflow[1] = 1;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    q_base[n_queries] = carry;
    if (carry > rec_cap) overflow[1] = 1;
  }
}

struct RecordArrays {
  u32 *qi;      // query descriptor index inside its query
  u32 *pos;     // sorted table position p
  u32 *frame;   // frame id of the entry
  unsigned char *cell;  // voxel_round index (diagnostic)
  double *dis;  // optional (diagnostic), may be null
};

// replays the mask words of every descriptor and writes its matches at the
// deterministic offset q_base[q] + rec_off[d] + rank: global (i, cell, j) order
__global__ __launch_bounds__(SGTD_PROBE_THREADS) void emit_kernel(TableView T, QueryView Q,
                                                                  ProbeBuffers B, const u32 *rec_off,
                                                                  const u32 *q_base, RecordArrays R) {
  constexpr int NW = SGTD_PROBE_THREADS / SGTD_WAVE;
  __shared__ u32 s_off[NW][32];
  __shared__ u32 s_start[NW][32];
  if (B.overflow[0] || B.overflow[1]) return;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const long long n_waves = (long long)gridDim.x * NW;
  const long long n_slots = (long long)Q.n_queries * Q.stride;
  for (long long d = (long long)blockIdx.x * NW + wid; d < n_slots; d += n_waves) {
    const int q = (int)(d / Q.stride);
    const u32 i = (u32)(d - (long long)q * Q.stride);
    if (i >= Q.count[q]) continue;
    if (B.n_match[d] == 0) continue;
    const double q0 = Q.side[d * 3 + 0], q1 = Q.side[d * 3 + 1], q2 = Q.side[d * 3 + 2];
    const u32 code = label_code(Q.label[d * 3 + 0], Q.label[d * 3 + 1], Q.label[d * 3 + 2]);
    u32 start, len;
    resolve_cells(T, q0, q1, q2, code, start, len);
    const u32 inc = wave_incl_scan(len);
    const u32 total = __shfl(inc, SGTD_WAVE - 1);
    if (lane < 32) {
      s_off[wid][lane] = (lane < SGTD_NCELL) ? inc - len : 0xFFFFFFFFu;
      s_start[wid][lane] = start;
    }
    __builtin_amdgcn_wave_barrier();
    const u32 n_words = (total + 63u) >> 6;
    const u32 mbase = B.mask_ptr[d];
    size_t out = (size_t)q_base[q] + rec_off[d];
    for (u32 w = 0; w < n_words; w++) {
      const u64 m = B.mask_words[mbase + w];
      if (m == 0) continue;
      if ((m >> lane) & 1ull) {
        int cell; u32 e;
        locate(s_off[wid], s_start[wid], (w << 6) + lane, cell, e);
        const size_t o = out + __popcll(m & lanemask_lt());
        R.qi[o] = i;
        R.pos[o] = e;
        R.frame[o] = T.frame[e];
        R.cell[o] = (unsigned char)cell;
        if (R.dis) R.dis[o] = norm3(q0 - T.s0[e], q1 - T.s1[e], q2 - T.s2[e]);
      }
      out += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// top candidate_num frames of one query (:423-433): repeated arg-max of
// (votes, lowest frame id), requires votes >= 5; marks slot_of[frame] = slot
__global__ __launch_bounds__(256) void topk_kernel(const u32 *votes_all, u32 frame_span, u32 frame_lo,
                                                   int cand_num, int *n_cand, int *cand_frame,
                                                   int *cand_votes, unsigned char *slot_of_all) {
  __shared__ u64 red[256 / SGTD_WAVE];
  __shared__ u64 picked[SGTD_MAX_CAND];
  __shared__ int n_picked;
  const int q = blockIdx.x;
  const u32 *votes = votes_all + (size_t)q * frame_span;
  unsigned char *slot_of = slot_of_all + (size_t)q * frame_span;
  if (threadIdx.x == 0) n_picked = 0;
  __syncthreads();
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
        picked[n_picked] = b;
        slot_of[f] = (unsigned char)n_picked;
        cand_frame[q * cand_num + n_picked] = (int)(f + frame_lo);
        cand_votes[q * cand_num + n_picked] = (int)v;
        n_picked++;
      } else {
        n_picked |= 0x40000000;  // stop marker
      }
    }
    __syncthreads();
    if (n_picked & 0x40000000) break;
    __threadfence_block();
  }
  if (threadIdx.x == 0) n_cand[q] = n_picked & 0x3FFFFFFF;
}

// stable split of a query's ordered rough matches by candidate slot:
// pair_off[q][k] offsets (relative to q_base[q]) and (q_idx, db_entry) pairs
__global__ __launch_bounds__(256) void assemble_kernel(const u32 *q_base, RecordArrays R,
                                                       const unsigned char *slot_of_all, u32 frame_span,
                                                       u32 frame_lo, const u32 *perm, int cand_num,
                                                       const int *n_cand, long long *pair_off,
                                                       u32 *pair_qi, u32 *pair_entry, const int *overflow) {
  constexpr int NW = 256 / SGTD_WAVE;
  __shared__ u32 run[64];
  __shared__ u32 wcount[NW][64];
  __shared__ u32 hist[64];
  if (overflow[0] || overflow[1]) return;
  const int q = blockIdx.x, tid = threadIdx.x, wid = tid >> 6;
  const unsigned char *slot_of = slot_of_all + (size_t)q * frame_span;
  const u32 lo = q_base[q], hi = q_base[q + 1];
  const int nc = n_cand[q];
  if (tid < 64) hist[tid] = 0;
  __syncthreads();
  // pass A: candidate histogram
  for (u32 r = lo + tid; r < hi; r += 256) {
    const unsigned char s = slot_of[R.frame[r] - frame_lo];
    if (s != 0xFF) atomicAdd(&hist[s], 1u);
  }
  __syncthreads();
  if (tid == 0) {
    u32 acc = 0;
    for (int k = 0; k < nc; k++) {
      run[k] = acc;
      pair_off[(size_t)q * (cand_num + 1) + k] = acc;
      acc += hist[k];
    }
    for (int k = nc; k <= cand_num; k++) pair_off[(size_t)q * (cand_num + 1) + k] = acc;
    for (int k = nc; k < 64; k++) run[k] = acc;
  }
  __syncthreads();
  // pass B: stable scatter, 256 records per round
  for (u32 r0 = lo; r0 < hi; r0 += 256) {
    if (tid < 64) {
#pragma unroll
      for (int w = 0; w < NW; w++) wcount[w][tid] = 0;
    }
    __syncthreads();
    const u32 r = r0 + tid;
    unsigned char s = 0xFF;
    if (r < hi) s = slot_of[R.frame[r] - frame_lo];
    const bool valid = s != 0xFF;
    u32 rank, count;
    wave_group_rank<6>((u32)s & 63u, valid, rank, count);
    if (valid && rank == 0) wcount[wid][s] = count;
    __syncthreads();
    if (valid) {
      u32 p = run[s] + rank;
#pragma unroll
      for (int w = 0; w < NW; w++)
        if (w < wid) p += wcount[w][s];
      pair_qi[(size_t)lo + p] = R.qi[r];
      pair_entry[(size_t)lo + p] = perm[R.pos[r]];
    }
    __syncthreads();
    if (tid < 64) {
      u32 add = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) add += wcount[w][tid];
      run[tid] += add;
    }
  }
}
