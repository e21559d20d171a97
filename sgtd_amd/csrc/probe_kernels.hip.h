// probe_kernels.hip.h — candidate_selector (src/sgtd/src/STDesc.cpp:318-460)
//
//   probe     (:351-400) work item = (query, chunk of 128 query descriptors),
//             dequeued by persistent workgroups; one wavefront per descriptor:
//             27 lanes resolve the 27 cells (truncating (int)(side+inc), gate
//             ||side-centre|| < 1.5, hash lookup key -> bucket), then all 64
//             lanes stream the concatenated bucket ranges coalesced from the
//             28-B/entry hot arrays.  Matches are compacted in (cell, j) order
//             by __ballot/popcount prefix into a per-descriptor list (frame,
//             entry), votes (:404-420) go to an LDS histogram per work item
//             that is flushed with one global atomic per touched frame.
//   topk      (:423-433) candidate_num rounds of arg-max over the votes:
//             votes desc, frame id asc, stop below 5 votes
//   assemble  (:434-449) per tile of 32 descriptors: count matches per
//             candidate slot, scan over tiles, then write every candidate's
//             match_list_ in the reference's (i, cell, j) order — a stable
//             split by slot done with wave_group_rank ballots
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
  u32 *rec_frame;       // [rec_cap] frame id of a match
  u32 *rec_entry;       // [rec_cap] sorted table position of a match
  unsigned char *rec_cell;  // [rec_cap] voxel_round index (diagnostic build only)
  double *rec_dis;      // [rec_cap] distance (diagnostic build only)
  u32 rec_cap;
  u32 *rec_cursor;      // global slab cursor
  u32 *item_cursor;     // work queue head
  u32 *list_ptr;        // [n_slots] first record of descriptor
  u32 *n_visit;         // [n_slots] entries visited by descriptor
  u32 *n_match;         // [n_slots] matches of descriptor
  u32 *votes;           // [n_queries * frame_span]
  int *overflow;        // [2]: 0 match records, 1 candidate pairs
};

#define SGTD_PROBE_THREADS 256
#define SGTD_PROBE_CHUNK 128    // query descriptors per work item
#define SGTD_REC_SLAB 2048u     // match records a wave takes from the global cursor at once
#define SGTD_TILE_DESCS 32      // query descriptors per assemble tile

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
__device__ __forceinline__ void locate(const u32 *cell_off /*[32] LDS*/, const u32 *cell_start,
                                       u32 pos, int &cell, u32 &entry) {
  // branch-free binary search for the last cell with off <= pos (offsets ascending,
  // entries 27..31 are UINT_MAX)
  int c = 0;
  if (cell_off[c + 16] <= pos) c += 16;
  if (cell_off[c + 8] <= pos) c += 8;
  if (cell_off[c + 4] <= pos) c += 4;
  if (cell_off[c + 2] <= pos) c += 2;
  if (cell_off[c + 1] <= pos) c += 1;
  cell = c;
  entry = cell_start[c] + (pos - cell_off[c]);
}

template <bool LDS_VOTES, bool DIAG>
__global__ __launch_bounds__(SGTD_PROBE_THREADS) void probe_kernel(TableView T, QueryView Q,
                                                                   ProbeBuffers B, double rough,
                                                                   int chunks_per_query) {
  constexpr int NW = SGTD_PROBE_THREADS / SGTD_WAVE;
  extern __shared__ u32 s_hist[];  // [frame_span] when LDS_VOTES
  __shared__ u32 s_off[NW][32];    // exclusive offsets, padded to 32 with UINT_MAX
  __shared__ u32 s_start[NW][32];
  __shared__ u32 s_item;
  const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
  const u32 n_items = (u32)Q.n_queries * (u32)chunks_per_query;
  u32 slab_next = 0, slab_end = 0;  // this wave's private record slab

  while (true) {
    if (tid == 0) s_item = atomicAdd(B.item_cursor, 1u);
    __syncthreads();
    const u32 item = s_item;
    if (item >= n_items) break;
    const int q = (int)(item / (u32)chunks_per_query);
    const u32 d_first = (item - (u32)q * (u32)chunks_per_query) * SGTD_PROBE_CHUNK;
    const u32 cnt = Q.count[q];
    if (d_first >= cnt) { __syncthreads(); continue; }
    const u32 d_last = min(d_first + SGTD_PROBE_CHUNK, cnt);
    if (LDS_VOTES) {
      for (u32 f = tid; f < T.frame_span; f += SGTD_PROBE_THREADS) s_hist[f] = 0;
    }
    __syncthreads();
    u32 *votes = B.votes + (size_t)q * T.frame_span;

    for (u32 i = d_first + wid; i < d_last; i += NW) {
      const long long d = (long long)q * Q.stride + i;
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
      // records of one descriptor are contiguous: make sure the slab can take
      // the worst case (every visited entry matches)
      if (total && slab_next + total > slab_end) {
        const u32 take = total > SGTD_REC_SLAB ? total : SGTD_REC_SLAB;
        u32 got = 0;
        if (lane == 0) got = atomicAdd(B.rec_cursor, take);
        got = __shfl(got, 0);
        slab_next = got; slab_end = got + take;
      }
      const bool fits = (unsigned long long)slab_next + total <= (unsigned long long)B.rec_cap;
      if (!fits && lane == 0) B.overflow[0] = 1;
      __builtin_amdgcn_wave_barrier();

      u32 matches = 0;
      const u32 n_words = (total + 63u) >> 6;
      for (u32 w = 0; w < n_words; w++) {
        const u32 pos = (w << 6) + lane;
        bool hit = false;
        u32 e = 0, fr = 0; int cell = 0; double dis = 0;
        if (pos < total) {
          locate(s_off[wid], s_start[wid], pos, cell, e);
          const double dx = q0 - T.s0[e], dy = q1 - T.s1[e], dz = q2 - T.s2[e];
          fr = T.frame[e];
          dis = norm3(dx, dy, dz);
          // unsigned (src.frame_id_ - db.frame_id_) > 0  <=>  ids differ (:373)
          hit = (qframe != fr) && (dis < thr);                    // :374-378
        }
        if (hit) {                                                // :410
          if (LDS_VOTES) atomicAdd(&s_hist[fr - T.frame_lo], 1u);
          else atomicAdd(&votes[fr - T.frame_lo], 1u);
        }
        const u64 m = __ballot(hit);
        if (hit && fits) {
          const u32 o = slab_next + matches + __popcll(m & lanemask_lt());
          B.rec_frame[o] = fr;
          B.rec_entry[o] = e;
          if (DIAG) { B.rec_cell[o] = (unsigned char)cell; B.rec_dis[o] = dis; }
        }
        matches += __popcll(m);
      }
      if (lane == 0) {
        B.list_ptr[d] = slab_next;
        B.n_visit[d] = total;
        B.n_match[d] = fits ? matches : 0;
      }
      if (fits) slab_next += matches;
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (LDS_VOTES) {
      for (u32 f = tid; f < T.frame_span; f += SGTD_PROBE_THREADS) {
        const u32 v = s_hist[f];
        if (v) atomicAdd(&votes[f], v);
      }
    }
  }
}

// top candidate_num frames of one query (:423-433): repeated arg-max of
// (votes, lowest frame id), requires votes >= 5; marks slot_of[frame] = slot
__global__ __launch_bounds__(256) void topk_kernel(const u32 *votes_all, u32 frame_span, u32 frame_lo,
                                                   int cand_num, int *n_cand, int *cand_frame,
                                                   int *cand_votes, unsigned char *slot_of_all) {
  __shared__ u64 red[256 / SGTD_WAVE];
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

// rows[dd][s] += matches of descriptor d_first+dd that belong to candidate slot s
__device__ __forceinline__ void tile_count_rows(const QueryView &Q, const ProbeBuffers &B,
                                                const unsigned char *slot_of, u32 frame_lo, int q,
                                                u32 d_first, u32 d_last, u32 (*rows)[64]) {
  constexpr int NW = 256 / SGTD_WAVE;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  for (u32 i = d_first + wid; i < d_last; i += NW) {
    const long long d = (long long)q * Q.stride + i;
    const u32 n = B.n_match[d], p0 = B.list_ptr[d], dd = i - d_first;
    for (u32 k0 = 0; k0 < n; k0 += SGTD_WAVE) {
      const u32 k = k0 + lane;
      unsigned char s = 0xFF;
      if (k < n) s = slot_of[B.rec_frame[p0 + k] - frame_lo];
      u32 rank, count;
      wave_group_rank<6>((u32)s & 63u, s != 0xFF, rank, count);
      if (s != 0xFF && rank == 0) rows[dd][s] += count;   // one lane per slot, row owned by this wave
    }
  }
}

// pass 1: tile_count[(q*tiles+tile)*64 + s] = matches of the tile in slot s;
// also the per-query sums of visited entries / matches for the statistics
__global__ __launch_bounds__(256) void tile_count_kernel(QueryView Q, ProbeBuffers B,
                                                         const unsigned char *slot_of_all, u32 frame_span,
                                                         u32 frame_lo, int tiles_per_query, u32 *tile_count,
                                                         u32 *q_M, unsigned long long *q_P) {
  __shared__ u32 rows[SGTD_TILE_DESCS][64];
  if (B.overflow[0]) return;
  const int q = blockIdx.x / tiles_per_query, tile = blockIdx.x % tiles_per_query;
  const int tid = threadIdx.x;
  const u32 cnt = Q.count[q];
  const u32 d_first = (u32)tile * SGTD_TILE_DESCS;
  u32 *out = tile_count + ((size_t)q * tiles_per_query + tile) * 64;
  if (d_first >= cnt) {
    if (tid < 64) out[tid] = 0;
    return;
  }
  const u32 d_last = min(d_first + SGTD_TILE_DESCS, cnt);
  for (int k = tid; k < SGTD_TILE_DESCS * 64; k += 256) (&rows[0][0])[k] = 0;
  __syncthreads();
  tile_count_rows(Q, B, slot_of_all + (size_t)q * frame_span, frame_lo, q, d_first, d_last, rows);
  __syncthreads();
  if (tid < 64) {
    u32 tot = 0;
    for (u32 dd = 0; dd < d_last - d_first; dd++) tot += rows[dd][tid];
    out[tid] = tot;
  }
  if (tid < (int)(d_last - d_first)) {
    const long long d = (long long)q * Q.stride + d_first + tid;
    atomicAdd(&q_M[q], B.n_match[d]);
    atomicAdd(&q_P[q], (unsigned long long)B.n_visit[d]);
  }
}

// per query: exclusive scan of tile_count over tiles (in place) per slot, then
// over slots: pair_off[q][k] (relative to the query's first pair), q_pairs[q]
__global__ __launch_bounds__(64) void tile_scan_kernel(u32 *tile_count, int tiles_per_query, int cand_num,
                                                       const int *n_cand, long long *pair_off, u32 *q_pairs,
                                                       const int *overflow) {
  __shared__ u32 tot[64];
  if (overflow[0]) return;
  const int q = blockIdx.x, s = threadIdx.x;
  u32 *tc = tile_count + (size_t)q * tiles_per_query * 64;
  u32 run = 0;
  int t = 0;
  for (; t + 8 <= tiles_per_query; t += 8) {
    u32 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = tc[(size_t)(t + k) * 64 + s];
#pragma unroll
    for (int k = 0; k < 8; k++) { tc[(size_t)(t + k) * 64 + s] = run; run += v[k]; }
  }
  for (; t < tiles_per_query; t++) { u32 v = tc[(size_t)t * 64 + s]; tc[(size_t)t * 64 + s] = run; run += v; }
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

// pass 2: every candidate's match_list_ in (i, cell, j) order (:437-449)
__global__ __launch_bounds__(256) void tile_write_kernel(QueryView Q, ProbeBuffers B,
                                                         const unsigned char *slot_of_all, u32 frame_span,
                                                         u32 frame_lo, const u32 *perm, int tiles_per_query,
                                                         const u32 *tile_excl, int cand_num,
                                                         const long long *pair_off, const u32 *q_pair_base,
                                                         u32 *pair_qi, u32 *pair_entry) {
  constexpr int NW = 256 / SGTD_WAVE;
  __shared__ u32 rows[SGTD_TILE_DESCS][64];
  if (B.overflow[0] || B.overflow[1]) return;
  const int q = blockIdx.x / tiles_per_query, tile = blockIdx.x % tiles_per_query;
  const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
  const u32 cnt = Q.count[q];
  const u32 d_first = (u32)tile * SGTD_TILE_DESCS;
  if (d_first >= cnt) return;
  const u32 d_last = min(d_first + SGTD_TILE_DESCS, cnt);
  const unsigned char *slot_of = slot_of_all + (size_t)q * frame_span;
  for (int k = tid; k < SGTD_TILE_DESCS * 64; k += 256) (&rows[0][0])[k] = 0;
  __syncthreads();
  tile_count_rows(Q, B, slot_of, frame_lo, q, d_first, d_last, rows);
  __syncthreads();
  if (tid < 64) {
    // counts -> absolute first output position of (descriptor, slot)
    u32 run = 0;
    if (tid < cand_num)
      run = q_pair_base[q] + (u32)pair_off[(size_t)q * (cand_num + 1) + tid] +
            tile_excl[((size_t)q * tiles_per_query + tile) * 64 + tid];
    for (u32 dd = 0; dd < d_last - d_first; dd++) {
      const u32 c = rows[dd][tid];
      rows[dd][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  for (u32 i = d_first + wid; i < d_last; i += NW) {
    const long long d = (long long)q * Q.stride + i;
    const u32 n = B.n_match[d], p0 = B.list_ptr[d], dd = i - d_first;
    for (u32 k0 = 0; k0 < n; k0 += SGTD_WAVE) {
      const u32 k = k0 + lane;
      unsigned char s = 0xFF;
      u32 e = 0;
      if (k < n) {
        s = slot_of[B.rec_frame[p0 + k] - frame_lo];
        e = B.rec_entry[p0 + k];
      }
      u32 rank, count;
      wave_group_rank<6>((u32)s & 63u, s != 0xFF, rank, count);
      u32 base = 0;
      if (s != 0xFF) base = rows[dd][s];
      __builtin_amdgcn_wave_barrier();
      if (s != 0xFF) {
        const u32 o = base + rank;
        pair_qi[o] = i;
        pair_entry[o] = perm[e];
        if (rank == 0) rows[dd][s] = base + count;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// diagnostic: the ordered rough-match list of ONE query (reference order i, cell, j)
__global__ __launch_bounds__(256) void rough_gather_kernel(QueryView Q, ProbeBuffers B, const u32 *perm, int q,
                                                           u32 *out_qi, u32 *out_entry, u32 *out_frame,
                                                           unsigned char *out_cell, double *out_dis) {
  __shared__ u32 lds[256 / SGTD_WAVE + 1];
  const u32 cnt = Q.count[q];
  u32 carry = 0;
  for (u32 i0 = 0; i0 < cnt; i0 += 256) {
    const u32 i = i0 + threadIdx.x;
    const long long d = (long long)q * Q.stride + i;
    const u32 n = (i < cnt) ? B.n_match[d] : 0;
    u32 tot;
    const u32 ex = block_excl_scan(n, lds, tot);
    if (i < cnt) {
      const u32 p0 = B.list_ptr[d];
      for (u32 k = 0; k < n; k++) {
        const u32 o = carry + ex + k;
        out_qi[o] = i;
        out_entry[o] = perm[B.rec_entry[p0 + k]];
        out_frame[o] = B.rec_frame[p0 + k];
        if (out_cell) out_cell[o] = B.rec_cell[p0 + k];
        if (out_dis) out_dis[o] = B.rec_dis[p0 + k];
      }
    }
    carry += tot;
  }
}
