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
//   slices     every bucket is partitioned (stably) into 2 x 3 sub-cells (halves of the
//              second side's cell interval x thirds of the third side's) + an overflow slice; entries of one frame
//              that could match the same query descriptor never straddle slices
//              (slice_assign_kernel), so per-frame match order stays the reference's
//   gather     probe layout in that order p: HotEntry {f32 sides, entry id} 16 B (+ perm[p] = g for the table dump)
//   csr+hash   bucket directory (start + slice counts) and open-addressing table key -> bucket
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

// non-monotone frame ids only: (frame, g) pairs, and the table key of every g of an order
__global__ void frame_keys_kernel(const u32 *frame, u64 *keys, u32 *vals, long long n) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  keys[g] = frame[g];
  vals[g] = (u32)g;
}
__global__ void keys_of_order_kernel(const double *side, const int *label, const u32 *order, u64 *keys, long long n) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t g = order[p];
  const int x = (int)(side[g * 3 + 0] + 0.5), y = (int)(side[g * 3 + 1] + 0.5), z = (int)(side[g * 3 + 2] + 0.5);
  keys[p] = pack_key(label_code(label[g * 3], label[g * 3 + 1], label[g * 3 + 2]), (u32)x & 65535u, (u32)y & 65535u, (u32)z & 65535u);
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

// hist[digit * nblocks + block]   (KeyT: u64, or u32 for keys of at most 32 bits — a third less traffic per pass)
template <class KeyT>
__global__ __launch_bounds__(SGTD_RS_THREADS) void radix_hist_kernel(const KeyT *keys, long long n, int shift,
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
template <class KeyT>
__global__ __launch_bounds__(SGTD_RS_THREADS) void radix_scatter_kernel(
    const KeyT *keys_in, const u32 *vals_in, KeyT *keys_out, u32 *vals_out, long long n,
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
    KeyT k = valid ? keys_in[i] : 0;
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
// side / frame are the arrays of the segment's range (indexed from its first entry g0); perm
// comes in with positions inside the range and leaves with insertion indices
__global__ void gather_hot_kernel(u32 *perm, const double *side, const u32 *frame, const u32 *id_of_g, IdMap map,
                                  HotEntry *ent, long long n, u32 g0) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) {
    // SGTD_SENTINELS entries behind the last one: where the sweep's lanes beyond a visit list
    // read (sides +inf: the f32 test rejects them)
    if (p < n + SGTD_SENTINELS) {
      HotEntry h;
      h.s0 = h.s1 = h.s2 = __builtin_inff();
      h.id = SGTD_DEAD_ID;
      ent[p] = h;
      perm[p] = 0;
    }
    return;
  }
  const u32 g = perm[p];         // position inside the segment's range of entries
  perm[p] = g + g0;
  HotEntry h;
  h.s0 = (float)side[(size_t)g * 3 + 0];   // round to nearest: the bound in f32_bounds assumes it
  h.s1 = (float)side[(size_t)g * 3 + 1];
  h.s2 = (float)side[(size_t)g * 3 + 2];
  if (id_of_g) h.id = id_of_g[g + g0];
  else {
    const u32 f = frame[g] - map.frame_lo;
    h.id = (f << map.bits) | ((g + g0) - map.frame_first[f]);
  }
  ent[p] = h;
}

// the frame runs of `key` (frame ids: in insertion order when they are monotone, else sorted):
// first[f - lo] = first position of frame f ...
__global__ void frame_first_kernel(const u32 *key, long long n, u32 lo, u32 *first) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  if (p == 0 || key[p - 1] != key[p]) first[key[p] - lo] = (u32)p;
}
// ... and the longest run (the largest number of entries of one frame)
__global__ void frame_longest_kernel(const u32 *key, long long n, u32 lo, const u32 *first, u32 *longest) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  if (p == n - 1 || key[p + 1] != key[p]) atomicMax(longest, (u32)(p + 1) - first[key[p] - lo]);
}
// frames out of insertion order: id of every entry from its position in the frame-major list
__global__ void id_of_sorted_kernel(const u32 *key, const u32 *by_frame, long long n, u32 lo, const u32 *first,
                                    u32 bits, u32 *id_of_g) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const u32 f = key[p] - lo;
  id_of_g[by_frame[p]] = (f << bits) | ((u32)p - first[f]);
}
__global__ void low_words_kernel(const u64 *keys, long long n, u32 *out) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) out[g] = (u32)keys[g];
}

// frame ids non-decreasing in insertion order?  (true for maps built frame by frame; then the
// stable sort by key alone leaves every bucket ordered by (frame, g))
__global__ void frame_monotone_kernel(const u32 *frame, long long n, int *flag) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g + 1 < n && frame[g + 1] < frame[g]) *flag = 1;
}

// sub-cell of an entry inside its cell: per axis the cell is (int)(s + 0.5) (STDesc.cpp:156-157),
// the slice the half / third of [cell, cell + 1) that s + 0.5 falls into.  (The product with 3
// is rounded: an entry within an ulp of a third's boundary may land on either side — the plan's
// slice bounds carry a margin of 1e-4 slice, table_kernels and sweep only have to agree on
// where the entry IS, which they do by construction: both read slice_of / the directory.)
__device__ __forceinline__ u32 axis_slice(double s, int n) {
  const double y = s + 0.5;
  const int cell = (int)y;
  const int sl = (int)(y * (double)n) - cell * n;
  return (u32)min(max(sl, 0), n - 1);
}
__device__ __forceinline__ u32 sub_cell(double s1, double s2) {
  return axis_slice(s1, SGTD_YSLICES) * SGTD_ZSLICES + axis_slice(s2, SGTD_ZSLICES);
}

// Slice assignment.  Input: entries sorted by (key, frame, g) (order[p] = g, keys[p]).  One
// thread per position; the head of every (key, frame) run decides for the run: if two of its
// members lie in different sub-cells AND could both match one query descriptor, the whole run
// goes to the overflow slice, otherwise every member keeps its sub-cell.  Two entries a, b can
// both match a query q only if ||a - b|| < 2 thr(q) and thr(q) = rough ||q|| <=
// rough (||a|| + thr(q)), i.e. thr(q) <= rough ||a|| / (1 - rough): the test below uses the
// larger norm and a relative margin.  Runs longer than SGTD_RUN_MAX members go to the
// overflow slice untested (always correct: that slice is in insertion order and always visited).
#define SGTD_RUN_MAX 48
__global__ void slice_assign_kernel(const u64 *keys, const u32 *order, const double *side, const u32 *frame,
                                    long long n, double rough, unsigned char *slice_of /*[g]*/) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const u32 g0 = order[p];
  const u64 key = keys[p];
  const u32 fr = frame[g0];
  if (p > 0 && keys[p - 1] == key && frame[order[p - 1]] == fr) return;   // not a run head
  long long e = p + 1;
  while (e < n && e - p <= SGTD_RUN_MAX && keys[e] == key && frame[order[e]] == fr) e++;
  const int len = (int)(e - p);
  if (len == 1) { slice_of[g0] = (unsigned char)sub_cell(side[(size_t)g0 * 3 + 1], side[(size_t)g0 * 3 + 2]); return; }
  bool overflow = len > SGTD_RUN_MAX;
  if (overflow) {   // mark the rest of the long run as well
    while (e < n && keys[e] == key && frame[order[e]] == fr) e++;
  } else {
    const double f = (rough < 1.0) ? 2.0 * rough / (1.0 - rough) * (1.0 + 1e-9) : __builtin_inf();
    for (int i = 0; i < len && !overflow; i++) {
      const u32 ga = order[p + i];
      const double a0 = side[(size_t)ga * 3], a1 = side[(size_t)ga * 3 + 1], a2 = side[(size_t)ga * 3 + 2];
      const u32 sa = sub_cell(a1, a2);
      const double na = norm3(a0, a1, a2);
      for (int j = i + 1; j < len; j++) {
        const u32 gb = order[p + j];
        const double b0 = side[(size_t)gb * 3], b1 = side[(size_t)gb * 3 + 1], b2 = side[(size_t)gb * 3 + 2];
        if (sub_cell(b1, b2) == sa) continue;
        const double nb = norm3(b0, b1, b2);
        const double lim = f * fmax(na, nb) + 1e-9;
        if (!(norm3(a0 - b0, a1 - b1, a2 - b2) > lim)) { overflow = true; break; }   // NaN counts as close
      }
    }
  }
  for (long long q = p; q < e; q++) {
    const u32 g = order[q];
    slice_of[g] = overflow ? (unsigned char)SGTD_NSUB : (unsigned char)sub_cell(side[(size_t)g * 3 + 1], side[(size_t)g * 3 + 2]);
  }
}

// Stable partition of every bucket by slice: one wavefront per bucket (grid-stride).  in[] holds
// the bucket's entries in insertion order (sorted by key, ties by g); out[] receives slice 0's
// entries, then slice 1's, ..., then the overflow slice's, each still in insertion order; the
// directory row of the bucket gets the cumulative slice counts.
__global__ __launch_bounds__(256) void slice_partition_kernel(const u32 *bucket_start, u32 n_buckets, u32 n_entries,
                                                              const u32 *in, const unsigned char *slice_of, u32 *out,
                                                              BucketDir *dir) {
  const int lane = lane_id();
  const u32 stride = (gridDim.x * blockDim.x) >> 6;
  for (u32 b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < n_buckets; b += stride) {
    const u32 start = bucket_start[b];
    const u32 end = (b + 1 < n_buckets) ? bucket_start[b + 1] : n_entries;
    u32 cnt[SGTD_NSLICE];
#pragma unroll
    for (int s = 0; s < SGTD_NSLICE; s++) cnt[s] = 0;
    for (u32 i = start + lane; i < end; i += SGTD_WAVE) {
      const u32 sl = slice_of[in[i]];
#pragma unroll
      for (int s = 0; s < SGTD_NSLICE; s++) cnt[s] += (sl == (u32)s) ? 1u : 0u;
    }
    u32 base[SGTD_NSLICE], acc = 0;
    BucketDir row;
    row.start = start;
#pragma unroll
    for (int s = 0; s < SGTD_NSLICE; s++) {
      base[s] = start + acc;
      acc += wave_sum(cnt[s]);
      row.cum[s] = acc;
    }
    if (lane == 0) dir[b] = row;
    for (u32 i0 = start; i0 < end; i0 += SGTD_WAVE) {
      const u32 i = i0 + lane;
      const bool valid = i < end;
      const u32 g = valid ? in[i] : 0u;
      const u32 sl = valid ? (u32)slice_of[g] : 0xFFu;
#pragma unroll
      for (int s = 0; s < SGTD_NSLICE; s++) {
        const u64 m = __ballot(sl == (u32)s);
        if (sl == (u32)s) out[base[s] + (u32)__popcll(m & lanemask_lt())] = g;
        base[s] += (u32)__popcll(m);
      }
    }
  }
}

// squared thresholds for caller-provided query descriptors
__global__ void thr2_kernel(const double *side, const u32 *frame, QueryRec *qrec, long long n, double rough) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  write_query_rec(qrec + i, side[i * 3], side[i * 3 + 1], side[i * 3 + 2], rough, frame[i]);
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
      table[h].bucket = b;
      table[h].len = end - start;
      return;
    }
    h = (h + 1) & mask;
  }
}

// sum over buckets of len^2 (a query descriptor distributed like the table's entries visits a
// bucket with probability len / E: sizes the first batch's work buffers, sgtd_accel.hip)
__global__ void bucket_sq_kernel(const u32 *bucket_start, u32 n_buckets, u32 n_entries, unsigned long long *sum) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = 0;
  if (b < n_buckets) {
    const unsigned long long len = ((b + 1 < n_buckets) ? bucket_start[b + 1] : n_entries) - bucket_start[b];
    v = len * len;
  }
  // wave reduction, one atomic per wave
#pragma unroll
  for (int d = SGTD_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
  if (lane_id() == 0 && v) atomicAdd(sum, v);
}

// sgtd_fetch_entries: table entries idx[0..n) gathered into contiguous staging arrays (one
// device-to-host copy per field afterwards instead of one per entry)
__device__ __forceinline__ void gather_entry(const long long *idx, long long i, const DescArrays &tab, const DescArrays &out, long long at = -1);
__global__ void gather_entries_kernel(const long long *idx, long long n, DescArrays tab, DescArrays out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  gather_entry(idx, i, tab, out);
}
// the same with the count still on the device (the launch covers an upper bound)
// (cap: the room of idx / out — the count on the device may exceed it, the caller then gathers again with more room)
__global__ void gather_entries_counted_kernel(const long long *idx, const long long *n_p, long long cap, DescArrays tab, DescArrays out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_p || i >= cap) return;
  gather_entry(idx, i, tab, out);
}
// the two in one launch (one frame per call): inlier pair i (q_idx << 32 | g) -> its query index and its table entry.  Eight lanes
// per pair, each with three or four words of the entry (the entry's seven arrays are seven scattered reads: spread over eight
// times the waves they overlap instead of queueing behind one another in one lane); the launch is a fixed number of
// workgroups that stride over the count, which is still on the device
// (`out` and `q_idx` may be page-locked HOST memory — sgtd_search_frame hands the caller's own arrays when the device can write them: the
// entries then cross the link once, inside this kernel, at the link's rate; members that are NULL are skipped)
// (`overflow`, or NULL: the batch's two flags — set: the pairs are not final and name nothing, nothing is read)
__global__ __launch_bounds__(256) void gather_pair_entries_kernel(const u64 *pairs, const long long *n_p, long long cap, int *q_idx, DescArrays tab, DescArrays out,
                                                                  const int *overflow) {
  if (overflow && (overflow[0] | overflow[1])) return;
  const long long n = *n_p < cap ? *n_p : cap;
  const int part = threadIdx.x & 7;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += ((long long)gridDim.x * blockDim.x) >> 3) {
    const u64 pr = pairs[i];
    const size_t g = (size_t)(pr & 0xFFFFFFFFull);
    switch (part) {
      case 0: if (out.side) for (int k = 0; k < 3; k++) out.side[i * 3 + k] = tab.side[g * 3 + k]; break;
      case 1: if (out.angle) for (int k = 0; k < 3; k++) out.angle[i * 3 + k] = tab.angle[g * 3 + k]; break;
      case 2: if (out.center) for (int k = 0; k < 3; k++) out.center[i * 3 + k] = tab.center[g * 3 + k]; break;
      case 3:
        if (out.label) for (int k = 0; k < 3; k++) out.label[i * 3 + k] = tab.label[g * 3 + k];
        if (out.frame) out.frame[i] = tab.frame[g];
        break;
      case 4:
        if (out.node_id) for (int k = 0; k < 3; k++) out.node_id[i * 3 + k] = tab.node_id[g * 3 + k];
        if (q_idx) q_idx[i] = (int)(pr >> 32);
        break;
      default:
        if (out.vertex) { const int k0 = (part - 5) * 3; for (int k = k0; k < k0 + 3; k++) out.vertex[i * 9 + k] = tab.vertex[g * 9 + k]; }
        break;
    }
  }
}
// pairs (q_idx << 32 | g) -> the two index lists
__global__ void split_pairs_kernel(const u64 *pairs, const long long *n_p, long long cap, long long *idx, int *q_idx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_p || i >= cap) return;
  const u64 pr = pairs[i];
  idx[i] = (long long)(pr & 0xFFFFFFFFull);
  q_idx[i] = (int)(pr >> 32);
}
__device__ __forceinline__ void gather_entry(const long long *idx, long long i, const DescArrays &tab, const DescArrays &out, long long at) {
  const size_t g = (size_t)idx[i];
  if (at >= 0) i = at;          // (entry idx[i] goes to slot `at` of the output)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    out.side[i * 3 + k] = tab.side[g * 3 + k];
    out.angle[i * 3 + k] = tab.angle[g * 3 + k];
    out.center[i * 3 + k] = tab.center[g * 3 + k];
    out.label[i * 3 + k] = tab.label[g * 3 + k];
    out.node_id[i * 3 + k] = tab.node_id[g * 3 + k];
  }
#pragma unroll
  for (int k = 0; k < 9; k++) out.vertex[i * 9 + k] = tab.vertex[g * 9 + k];
  out.frame[i] = tab.frame[g];
}

// descriptors that arrived as ONE block (side | angle | center | vertex | label | frame | node_id, `n` entries each, every field on
// a 16-byte boundary: desc_block_offsets) to the store's seven arrays from entry `first` on, word by word (every field is a
// multiple of four bytes).  `present` bit f: field f is in the block (an absent field was zeroed by a memset).
struct DescBlockOffsets { u32 off[7]; u32 total; };
__host__ __device__ __forceinline__ DescBlockOffsets desc_block_offsets(size_t n) {
  const u32 bytes[7] = {24, 24, 24, 36, 12, 4, 12};
  DescBlockOffsets o;
  u32 at = 0;
  for (int f = 0; f < 7; f++) { o.off[f] = at; at += (u32)((n * bytes[f] + 15) & ~(size_t)15); }
  o.total = at;
  return o;
}
__global__ void unpack_desc_block_kernel(const unsigned char *block, u32 n, u32 present, DescArrays out, long long first) {
  const DescBlockOffsets o = desc_block_offsets(n);
  u32 *dst[7] = {reinterpret_cast<u32 *>(out.side + first * 3), reinterpret_cast<u32 *>(out.angle + first * 3), reinterpret_cast<u32 *>(out.center + first * 3),
                 reinterpret_cast<u32 *>(out.vertex + first * 9), reinterpret_cast<u32 *>(out.label + first * 3), reinterpret_cast<u32 *>(out.frame + first),
                 reinterpret_cast<u32 *>(out.node_id + first * 3)};
  const u32 words[7] = {6, 6, 6, 9, 3, 1, 3};
  const u32 per = 34u;     // words of one entry over all fields
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n * per; i += gridDim.x * blockDim.x) {
    // word i of the block's payload, field by field
    u32 r = i, f = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) { const u32 span = n * words[k]; if (f == (u32)k && r >= span) { r -= span; f = (u32)k + 1; } }
    if ((present >> f) & 1u) dst[f][r] = reinterpret_cast<const u32 *>(block + o.off[f])[r];
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
