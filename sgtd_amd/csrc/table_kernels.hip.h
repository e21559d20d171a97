// table_kernels.hip.h — AddSTDescs (src/sgtd/src/STDesc.cpp:149-172) as a
// device-side geometric hash table build:
//
//   append     descriptors land in insertion order g (frame asc, descriptor
//              order inside a frame) — the order the reference's bucket
//              vectors grow in
//   key        cell = (int)(side + 0.5) per side (:155-157), 12-bit label code
//              (:161), packed as code|x|y|z (60 bits)
//   sort       stable LSD radix sort of (key, g) by key, 8-bit digits, passes
//              whose digit is constant are skipped; stable => inside a bucket
//              entries stay in insertion order == the reference's index j
//   gather     probe layout in sorted order p, two 16-B arrays: head {side0, side1 f64},
//              tail {side2 f64, frame u32, g u32} = 32 B per entry (28 algorithmic + g)
//   csr+hash   bucket boundaries -> open-addressing table key -> (start,len)
#pragma once
#include "common.hip.h"

#define SGTD_SCAN_THREADS 256
#define SGTD_SCAN_ITEMS 8   // per thread => 2048 per block

// ---------------------------------------------------------------------------
// device-wide exclusive scan of u32 (three-kernel, recursive on block sums)
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *lds_wave /*[>=waves+1]*/, u32 &block_total) {
  const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  u32 inc = wave_incl_scan(v);
  if (lane == SGTD_WAVE - 1) lds_wave[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    u32 w = (lane < nw) ? lds_wave[lane] : 0;
    u32 wi = wave_incl_scan(w);
    if (lane < nw) lds_wave[lane] = wi - w;
    if (lane == nw - 1) lds_wave[nw] = wi;
  }
  __syncthreads();
  u32 r = lds_wave[wid] + inc - v;
  block_total = lds_wave[nw];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(SGTD_SCAN_THREADS) void scan_reduce_kernel(const u32 *in, u32 *block_sums, long long n) {
  __shared__ u32 lds[SGTD_SCAN_THREADS / SGTD_WAVE + 1];
  long long base = (long long)blockIdx.x * SGTD_SCAN_THREADS * SGTD_SCAN_ITEMS;
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SGTD_SCAN_ITEMS; k++) {
    long long i = base + (long long)k * SGTD_SCAN_THREADS + threadIdx.x;
    if (i < n) s += in[i];
  }
  u32 tot;
  block_excl_scan(s, lds, tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// out[i] = offsets[block] + exclusive prefix inside the block; in-place allowed
__global__ __launch_bounds__(SGTD_SCAN_THREADS) void scan_apply_kernel(const u32 *in, u32 *out, const u32 *block_off, long long n) {
  __shared__ u32 lds[SGTD_SCAN_THREADS / SGTD_WAVE + 1];
  long long base = (long long)blockIdx.x * SGTD_SCAN_THREADS * SGTD_SCAN_ITEMS;
  // thread owns SGTD_SCAN_ITEMS consecutive items
  u32 v[SGTD_SCAN_ITEMS];
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SGTD_SCAN_ITEMS; k++) {
    long long i = base + (long long)threadIdx.x * SGTD_SCAN_ITEMS + k;
    v[k] = (i < n) ? in[i] : 0;
    s += v[k];
  }
  u32 tot;
  u32 ex = block_excl_scan(s, lds, tot) + (block_off ? block_off[blockIdx.x] : 0);
#pragma unroll
  for (int k = 0; k < SGTD_SCAN_ITEMS; k++) {
    long long i = base + (long long)threadIdx.x * SGTD_SCAN_ITEMS + k;
    if (i < n) out[i] = ex;
    ex += v[k];
  }
}

// ---------------------------------------------------------------------------
// keys
// ---------------------------------------------------------------------------
__global__ void make_keys_kernel(const double *side, const int *label, u64 *keys, u32 *vals,
                                 long long n, int *bad_flag) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  int x = (int)(side[g * 3 + 0] + 0.5), y = (int)(side[g * 3 + 1] + 0.5),
      z = (int)(side[g * 3 + 2] + 0.5);
  if ((unsigned)x > 65535u || (unsigned)y > 65535u || (unsigned)z > 65535u) {
    *bad_flag = 1;
    x &= 65535; y &= 65535; z &= 65535;
  }
  keys[g] = pack_key(label_code(label[g * 3], label[g * 3 + 1], label[g * 3 + 2]), (u32)x, (u32)y, (u32)z);
  vals[g] = (u32)g;
}

// a loaded table's frame ids must lie in the header's [lo, hi] (the votes are indexed by them)
__global__ void frame_range_check_kernel(const u32 *frame, long long n, u32 lo, u32 hi, int *bad_flag) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n && (frame[g] < lo || frame[g] > hi)) *bad_flag = 1;
}

// ---------------------------------------------------------------------------
// LSD radix sort, 8-bit digit per pass
// ---------------------------------------------------------------------------
#define SGTD_RS_THREADS 256
#define SGTD_RS_ROUNDS 16
#define SGTD_RS_TILE (SGTD_RS_THREADS * SGTD_RS_ROUNDS)

// hist[digit * nblocks + block]
__global__ __launch_bounds__(SGTD_RS_THREADS) void radix_hist_kernel(const u64 *keys, long long n, int shift,
                                                                      u32 *hist, int nblocks,
                                                                      const u32 *n_dev = nullptr) {
  __shared__ u32 h[256];
  if (n_dev) n = min(n, (long long)*n_dev);   // element count known only on the device
  h[threadIdx.x] = 0;
  __syncthreads();
  long long base = (long long)blockIdx.x * SGTD_RS_TILE;
#pragma unroll 4
  for (int r = 0; r < SGTD_RS_ROUNDS; r++) {
    long long i = base + (long long)r * SGTD_RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(u32)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// digit totals (to skip constant-digit passes): tot[d] = sum over blocks
__global__ void radix_digit_totals_kernel(const u32 *hist, int nblocks, u32 *tot) {
  __shared__ u32 lds[SGTD_SCAN_THREADS / SGTD_WAVE + 1];
  const int d = blockIdx.x;
  u32 s = 0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += hist[(size_t)d * nblocks + b];
  u32 t;
  block_excl_scan(s, lds, t);
  if (threadIdx.x == 0) tot[d] = t;
}

// stable scatter: hist now holds the exclusive scan (global base per digit,block)
__global__ __launch_bounds__(SGTD_RS_THREADS) void radix_scatter_kernel(
    const u64 *keys_in, const u32 *vals_in, u64 *keys_out, u32 *vals_out, long long n,
    int shift, const u32 *hist_scanned, int nblocks, const u32 *n_dev = nullptr) {
  constexpr int NW = SGTD_RS_THREADS / SGTD_WAVE;
  __shared__ u32 run[256];          // next free position per digit for this block
  __shared__ u32 wcount[NW][256];   // per-wave digit counts of the current round
  const int tid = threadIdx.x, wid = tid >> 6;
  if (n_dev) n = min(n, (long long)*n_dev);
  if ((long long)blockIdx.x * SGTD_RS_TILE >= n) return;   // uniform: nothing of this tile is live
  run[tid] = hist_scanned[(size_t)tid * nblocks + blockIdx.x];
  long long base = (long long)blockIdx.x * SGTD_RS_TILE;
  for (int r = 0; r < SGTD_RS_ROUNDS; r++) {
#pragma unroll
    for (int w = 0; w < NW; w++) wcount[w][tid] = 0;
    __syncthreads();
    long long i = base + (long long)r * SGTD_RS_THREADS + tid;
    const bool valid = i < n;
    u64 k = valid ? keys_in[i] : 0;
    u32 v = valid ? vals_in[i] : 0;
    u32 digit = (u32)(k >> shift) & 255u;
    u32 rank, count;
    wave_group_rank<8>(digit, valid, rank, count);
    if (valid && rank == 0) wcount[wid][digit] = count;
    __syncthreads();
    if (valid) {
      u32 pos = run[digit] + rank;
#pragma unroll
      for (int w = 0; w < NW; w++)
        if (w < wid) pos += wcount[w][digit];
      keys_out[pos] = k;
      vals_out[pos] = v;
    }
    __syncthreads();
    u32 add = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) add += wcount[w][tid];
    run[tid] += add;
    // (the zeroing at the top of the next round is ordered by the barrier above)
  }
}

// ---------------------------------------------------------------------------
// gather hot arrays + bucket heads + hash insert
// ---------------------------------------------------------------------------
__global__ void gather_hot_kernel(const u32 *perm, const double *side, const u32 *frame,
                                  HotHead *head, HotTail *tail, long long n) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const u32 g = perm[p];
  HotHead h;
  HotTail t;
  h.s0 = side[(size_t)g * 3 + 0];
  h.s1 = side[(size_t)g * 3 + 1];
  t.s2 = side[(size_t)g * 3 + 2];
  t.frame = frame[g];
  t.g = g;
  head[p] = h;
  tail[p] = t;
}

// squared thresholds for caller-provided query descriptors
__global__ void thr2_kernel(const double *side, const u32 *frame, QueryRec *qrec, long long n, double rough) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double s0 = side[i * 3], s1 = side[i * 3 + 1], s2 = side[i * 3 + 2];
  double2 *qr = reinterpret_cast<double2 *>(qrec + i);
  qr[0] = make_double2(s0, s1);
  qr[1] = make_double2(s2, sq_threshold(norm3(s0, s1, s2) * rough));
  reinterpret_cast<uint4 *>(qr)[2] = make_uint4(frame[i], gate_mask(s0, s1, s2), 0u, 0u);
}

__global__ void head_flags_kernel(const u64 *keys, u32 *flags, long long n) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  flags[p] = (p == 0 || keys[p] != keys[p - 1]) ? 1u : 0u;
}

// bucket_start[b] = p for the head p of bucket b (bid = exclusive scan of flags)
__global__ void bucket_starts_kernel(const u64 *keys, const u32 *bid_excl, u32 *bucket_start,
                                     u64 *bucket_key, long long n) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  if (p == 0 || keys[p] != keys[p - 1]) {
    bucket_start[bid_excl[p]] = (u32)p;
    bucket_key[bid_excl[p]] = keys[p];
  }
}

__global__ void hash_insert_kernel(const u64 *bucket_key, const u32 *bucket_start, u32 n_buckets,
                                   u32 n_entries, HashSlot *table, u32 mask) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_buckets) return;
  const u64 key = bucket_key[b];
  const u32 start = bucket_start[b];
  const u32 end = (b + 1 < n_buckets) ? bucket_start[b + 1] : n_entries;
  u32 h = hash_key(key) & mask;
  while (true) {
    u64 prev = atomicCAS(reinterpret_cast<u64 *>(&table[h].key), SGTD_EMPTY_KEY, key);
    if (prev == SGTD_EMPTY_KEY) {
      table[h].start = start;
      table[h].len = end - start;
      return;
    }
    h = (h + 1) & mask;
  }
}

// scatter of a strided build result into the table's cold arrays (append):
// frame k's count[k] descriptors go to g = gbase[k] + r
struct AppendParams {
  long long in_stride;
  const u32 *count;   // [n_frames]
  const u32 *goff;    // [n_frames] exclusive scan of count
  long long g0;       // entries before this batch
};
__global__ void append_descs_kernel(AppendParams A, DescArrays in, DescArrays tab) {
  const int f = blockIdx.x;
  const u32 cnt = A.count[f];
  const size_t src0 = (size_t)f * A.in_stride, dst0 = (size_t)A.g0 + A.goff[f];
  for (u32 r = threadIdx.x; r < cnt; r += blockDim.x) {
    const size_t s = src0 + r, d = dst0 + r;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      tab.side[d * 3 + k] = in.side[s * 3 + k];
      tab.angle[d * 3 + k] = in.angle[s * 3 + k];
      tab.center[d * 3 + k] = in.center[s * 3 + k];
      tab.label[d * 3 + k] = in.label[s * 3 + k];
      tab.node_id[d * 3 + k] = in.node_id[s * 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) tab.vertex[d * 9 + k] = in.vertex[s * 9 + k];
    tab.frame[d] = in.frame[s];
  }
}
