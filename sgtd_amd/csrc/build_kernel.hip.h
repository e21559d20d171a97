// build_kernel.hip.h — BuildSingleScanSTD (src/sgtd/src/STDesc.cpp:174-315)
// for a batch of frames, one workgroup per frame.
//
//   stage 0  keypoints (xyz f32 + label) -> LDS                (16 B/keypoint)
//   stage 1  exact k-NN in LDS: f32 ((dx*dx)+dy*dy)+dz*dz, ascending, ties ->
//            lower index (stands in for pcl::KdTreeFLANN, :183-192)
//   stage 2  triplet enumeration t = i*C(K-1,2) + (m,n) in the reference's
//            loop order (:189-194): sides, length filter (:204-208), 3-step
//            strict-'>' sort (:220-243), millimetre key (:244-248)
//   stage 3  first-wins dedup (:249-251,307): open-addressing table in LDS
//            holding the minimum t per key (atomicCAS + atomicMin)
//   stage 4  ordered compaction (winner bitmap + prefix popcount) and
//            descriptor fill (:253-308) at out[frame_slot*stride + rank]
#pragma once
#include "common.hip.h"

struct DescArrays {
  double *side, *angle, *center;  // [cap*3]
  float *vertex;                  // [cap*9]  A,B,C xyz
  int *label;                     // [cap*3]
  u32 *frame;                     // [cap]
  int *node_id;                   // [cap*3]
  QueryRec *qrec;                 // [cap] sweep record (query descriptors only; may be null)
};

#define SGTD_BUILD_THREADS 1024
#ifdef SGTD_EXP_PHASE
__device__ unsigned long long g_bphase[8];
#define BPH(i) do { if (tid == 0) { const unsigned long long _n = __builtin_readcyclecounter(); atomicAdd(&g_bphase[i], _n - bph_t); bph_t = _n; } } while (0)
#else
#define BPH(i) do { } while (0)
#endif
#define SGTD_TRI_INVALID 0xFFFFFFFFFFFFFFFFull
#define SGTD_SLOT_EMPTY 0xFFFFFFFFu

struct Tri {
  double a, b, c;     // sorted ascending
  int va, vb, vc;     // vertex A,B,C as 0/1/2 = p1/p2/p3
  bool valid;
};

// |p - q| as the reference computes it (:198-203): f32 subtraction, then
// pow(double,2) (exact), two rounded f64 adds, correctly rounded sqrt
__device__ __forceinline__ double side_len(const float4 &p, const float4 &q) {
  double dx = (double)(p.x - q.x), dy = (double)(p.y - q.y), dz = (double)(p.z - q.z);
  return sqrt(dx * dx + dy * dy + dz * dz);
}

__device__ __forceinline__ Tri make_triangle(const float4 &p1, const float4 &p2,
                                             const float4 &p3, const DevCfg &cfg) {
  Tri t;
  double a = side_len(p1, p2), b = side_len(p1, p3), c = side_len(p3, p2);
  t.valid = !(a > cfg.max_len || b > cfg.max_len || c > cfg.max_len ||
              a < cfg.min_len || b < cfg.min_len || c < cfg.min_len);
  // side identities 0:(p1,p2) 1:(p1,p3) 2:(p2,p3) stand for l1,l2,l3 (:213-218);
  // two different sides share vertex index (u+v-1): {0,1}->p1 {0,2}->p2 {1,2}->p3
  int sa = 0, sb = 1, sc = 2;
  double tmp; int ti;
  if (a > b) { tmp = a; a = b; b = tmp; ti = sa; sa = sb; sb = ti; }
  if (b > c) { tmp = b; b = c; c = tmp; ti = sb; sb = sc; sc = ti; }
  if (a > b) { tmp = a; a = b; b = tmp; ti = sa; sa = sb; sb = ti; }
  t.a = a; t.b = b; t.c = c;
  t.va = sa + sb - 1;  // shared by shortest and middle side  (:253-265)
  t.vb = sa + sc - 1;  // shared by shortest and longest      (:266-278)
  t.vc = sb + sc - 1;  // shared by middle and longest        (:279-291)
  return t;
}

// millimetre dedup key: (int64_t)(float)(side*1000) per side (:244-248), 21 bits each
__host__ __device__ __forceinline__ u64 pack_milli_key(u64 x, u64 y, u64 z) { return (x << 42) | (y << 21) | z; }
__device__ __forceinline__ u64 milli_key(const Tri &t) {
  float kx = (float)(t.a * 1000.0), ky = (float)(t.b * 1000.0), kz = (float)(t.c * 1000.0);
  u64 x = (u64)(long long)kx, y = (u64)(long long)ky, z = (u64)(long long)kz;
  return pack_milli_key(x, y, z);
}

struct BuildParams {
  const float *xyz;        // [total_kp*3]
  const u32 *label;        // [total_kp]
  const long long *kp_off; // [n_frames+1] device
  int n_frames;
  u32 frame_id0;
  int frame_id_step;       // 1: frame k gets frame_id0+k (map), 0: all frame_id0 (queries)
  long long out_stride;    // descriptor slots per frame
  u32 *out_count;          // [n_frames]
  int max_n;               // largest keypoint count in the batch (LDS layout)
  int max_slots;           // pow2 >= max_n*tpi + 1
  // global fallback for the dedup tables (LDS_DEDUP == false): per block
  u64 *ws_keys;            // [gridDim.x * max_n*tpi]
  u32 *ws_slots;           // [gridDim.x * max_slots]
};

// KM = slots of the k-NN insertion network, the smallest of {4, 8, 10, 12, 16} that holds
// descriptor_near_num (the shipped configuration is 10: a 16-slot network would do 60 % more work)
template <bool LDS_DEDUP, int KM>
__global__ __launch_bounds__(SGTD_BUILD_THREADS) void build_frames_kernel(
    BuildParams P, DevCfg cfg, DescArrays out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int K = cfg.K, tpi = cfg.tpi;
  const int max_t = P.max_n * tpi;
  const int max_words = (max_t + 31) / 32;
  // LDS carve (all offsets multiples of 16 bytes)
  float4 *pts = reinterpret_cast<float4 *>(smem);
  size_t off = (size_t)P.max_n * 16;
  unsigned short *knn = reinterpret_cast<unsigned short *>(smem + off);
  off += (((size_t)P.max_n * K * 2) + 15) & ~(size_t)15;
  unsigned char *mn = smem + off;
  off += (((size_t)tpi * 2) + 15) & ~(size_t)15;
  u32 *winbits = reinterpret_cast<u32 *>(smem + off);
  off += (((size_t)max_words * 4) + 15) & ~(size_t)15;
  u32 *wordpre = reinterpret_cast<u32 *>(smem + off);
  off += (((size_t)(max_words + 1) * 4) + 15) & ~(size_t)15;
  u64 *keys;
  u32 *slots;
  if (LDS_DEDUP) {
    keys = reinterpret_cast<u64 *>(smem + off);
    off += (size_t)max_t * 8;
    slots = reinterpret_cast<u32 *>(smem + off);
  } else {
    keys = P.ws_keys + (size_t)blockIdx.x * max_t;
    slots = P.ws_slots + (size_t)blockIdx.x * P.max_slots;
  }
  const int tid = threadIdx.x;

  // (m,n) rank pairs in loop order m = 1..K-2, n = m+1..K-1 (:193-194)
  if (tid == 0) {
    int k = 0;
    for (int m = 1; m < K - 1; m++)
      for (int n = m + 1; n < K; n++) {
        mn[2 * k] = (unsigned char)m;
        mn[2 * k + 1] = (unsigned char)n;
        k++;
      }
  }

  for (int f = blockIdx.x; f < P.n_frames; f += gridDim.x) {
    const long long kp0 = P.kp_off[f];
    const int n = (int)(P.kp_off[f + 1] - kp0);
    __syncthreads();
    if (n < K) {  // the reference indexes K neighbours regardless (UB): no descriptors
      if (tid == 0) P.out_count[f] = 0;
      continue;
    }
    const int T = n * tpi;
    int nslots = 64;
    while (nslots < T + 1) nslots <<= 1;
    const int nwords = (T + 31) / 32;

#ifdef SGTD_EXP_PHASE
    unsigned long long bph_t = __builtin_readcyclecounter();
#endif
    // ---- stage 0: keypoints -> LDS
    for (int i = tid; i < n; i += SGTD_BUILD_THREADS) {
      float4 p;
      p.x = P.xyz[(kp0 + i) * 3 + 0];
      p.y = P.xyz[(kp0 + i) * 3 + 1];
      p.z = P.xyz[(kp0 + i) * 3 + 2];
      p.w = __uint_as_float(P.label[kp0 + i]);
      pts[i] = p;
    }
    for (int s = tid; s < nslots; s += SGTD_BUILD_THREADS) slots[s] = SGTD_SLOT_EMPTY;
    for (int w = tid; w < nwords; w += SGTD_BUILD_THREADS) winbits[w] = 0;
    __syncthreads();

    BPH(0);
    // ---- stage 1: k-NN by sorted insertion networks in registers.  Four threads
    // share a keypoint: each scans a quarter of the candidates (ascending index),
    // parts 1..3 park their K best in the (not yet live) dedup-key area and part 0
    // inserts them in part order — the same sequence of insertions as one thread
    // scanning all candidates in index order, so ties still go to the lower index.
    const int parts = (3 * K <= tpi) ? 4 : 1;   // the parking area is T*8 bytes = n*tpi*8
    const int per_pass = SGTD_BUILD_THREADS / parts;
    for (int i0 = 0; i0 < n; i0 += per_pass) {
      const int i = i0 + tid / parts, part = tid % parts;
      float bd[KM];
      int bi[KM];
#pragma unroll
      for (int k = 0; k < KM; k++) { bd[k] = __builtin_inff(); bi[k] = 0; }
      auto insert = [&](float cd, int cj) {
        bool lt = false;
#pragma unroll
        for (int k = 0; k < KM; k++) {
          // strict '<': equal distance keeps the lower index first; once the new
          // point is placed every later slot shifts down by one (plain insertion)
          lt = lt || (cd < bd[k]);
          float td = lt ? bd[k] : cd; int tj = lt ? bi[k] : cj;
          bd[k] = lt ? cd : bd[k]; bi[k] = lt ? cj : bi[k];
          cd = td; cj = tj;
        }
      };
      if (i < n) {
        const float4 q = pts[i];
        const int j_lo = (int)((long long)n * part / parts), j_hi = (int)((long long)n * (part + 1) / parts);
        for (int j = j_lo; j < j_hi; j++) {
          const float4 p = pts[j];
          float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
          float d = dx * dx;   // FLANN L2_Simple accumulation order
          d += dy * dy;
          d += dz * dz;
          insert(d, j);
        }
        if (part > 0) {
#pragma unroll
          for (int k = 0; k < KM; k++)
            if (k < K) keys[((size_t)i * 3 + (part - 1)) * K + k] = ((u64)__float_as_uint(bd[k]) << 32) | (u32)bi[k];
        }
      }
      if (parts > 1) {
        if (!LDS_DEDUP) __threadfence_block();
        __syncthreads();
        if (i < n && part == 0) {
          for (int pk = 0; pk < 3 * K; pk++) {   // part 1, 2, 3, each in its own sorted order
            const u64 v = keys[(size_t)i * 3 * K + pk];
            insert(__uint_as_float((u32)(v >> 32)), (int)(u32)v);
          }
        }
      }
      if (i < n && part == 0) {
#pragma unroll
        for (int k = 0; k < KM; k++)
          if (k < K) knn[i * K + k] = (unsigned short)bi[k];
      }
      if (parts > 1) __syncthreads();   // the parking area is reused by the next pass
    }
    __syncthreads();

    BPH(1);
    // ---- stage 2: keys of all triplets
    for (int t = tid; t < T; t += SGTD_BUILD_THREADS) {
      const int i = t / tpi, r = t - i * tpi;
      const int m = mn[2 * r], nn = mn[2 * r + 1];
      Tri tr = make_triangle(pts[i], pts[knn[i * K + m]], pts[knn[i * K + nn]], cfg);
      keys[t] = tr.valid ? milli_key(tr) : SGTD_TRI_INVALID;
    }
    if (!LDS_DEDUP) __threadfence_block();
    __syncthreads();

    BPH(2);
    // ---- stage 3: first-wins dedup, slot value = min t of its key
    for (int t = tid; t < T; t += SGTD_BUILD_THREADS) {
      const u64 key = keys[t];
      if (key == SGTD_TRI_INVALID) continue;
      u32 h = (u32)mix64(key) & (u32)(nslots - 1);
      while (true) {
        u32 cur = atomicCAS(&slots[h], SGTD_SLOT_EMPTY, (u32)t);
        if (cur == SGTD_SLOT_EMPTY) break;
        if (keys[cur] == key) { atomicMin(&slots[h], (u32)t); break; }
        h = (h + 1) & (u32)(nslots - 1);
      }
    }
    __syncthreads();
    for (int s = tid; s < nslots; s += SGTD_BUILD_THREADS) {
      u32 v = slots[s];
      if (v != SGTD_SLOT_EMPTY) atomicOr(&winbits[v >> 5], 1u << (v & 31));
    }
    __syncthreads();
    // exclusive prefix of popcounts over the winner bitmap (wave 0)
    if (tid < SGTD_WAVE) {
      u32 carry = 0;
      for (int w0 = 0; w0 < nwords; w0 += SGTD_WAVE) {
        int w = w0 + tid;
        u32 c = (w < nwords) ? __popc(winbits[w]) : 0;
        u32 inc = wave_incl_scan(c);
        if (w < nwords) wordpre[w] = carry + inc - c;
        carry += __shfl(inc, SGTD_WAVE - 1);
      }
      if (tid == 0) { wordpre[nwords] = carry; P.out_count[f] = carry; }
    }
    __syncthreads();

    BPH(3);
    // ---- stage 4: descriptor fill in (i,m,n) order of the surviving triplets
    const u32 frame_id = P.frame_id0 + (u32)(P.frame_id_step * f);
    for (int t = tid; t < T; t += SGTD_BUILD_THREADS) {
      const u32 word = winbits[t >> 5];
      if (!((word >> (t & 31)) & 1u)) continue;
      const u32 rank = wordpre[t >> 5] + __popc(word & ((1u << (t & 31)) - 1u));
      const int i = t / tpi, r = t - i * tpi;
      const int m = mn[2 * r], nn = mn[2 * r + 1];
      float4 p[3];
      p[0] = pts[i]; p[1] = pts[knn[i * K + m]]; p[2] = pts[knn[i * K + nn]];
      const Tri tr = make_triangle(p[0], p[1], p[2], cfg);
      const float4 A = tr.va == 0 ? p[0] : (tr.va == 1 ? p[1] : p[2]);
      const float4 B = tr.vb == 0 ? p[0] : (tr.vb == 1 ? p[1] : p[2]);
      const float4 C = tr.vc == 0 ? p[0] : (tr.vc == 1 ? p[1] : p[2]);
      const size_t o = (size_t)f * P.out_stride + rank;
      const double a = tr.a, b = tr.b, c = tr.c;
      out.side[o * 3 + 0] = cfg.scale * a;
      out.side[o * 3 + 1] = cfg.scale * b;
      out.side[o * 3 + 2] = cfg.scale * c;
      out.angle[o * 3 + 0] = fabs((b * b + c * c - a * a) / (2 * b * c));  // :299-301
      out.angle[o * 3 + 1] = fabs((a * a + c * c - b * b) / (2 * a * c));
      out.angle[o * 3 + 2] = fabs((a * a + b * b - c * c) / (2 * a * b));
      out.center[o * 3 + 0] = (((double)A.x + (double)B.x) + (double)C.x) / 3;  // :296
      out.center[o * 3 + 1] = (((double)A.y + (double)B.y) + (double)C.y) / 3;
      out.center[o * 3 + 2] = (((double)A.z + (double)B.z) + (double)C.z) / 3;
      out.vertex[o * 9 + 0] = A.x; out.vertex[o * 9 + 1] = A.y; out.vertex[o * 9 + 2] = A.z;
      out.vertex[o * 9 + 3] = B.x; out.vertex[o * 9 + 4] = B.y; out.vertex[o * 9 + 5] = B.z;
      out.vertex[o * 9 + 6] = C.x; out.vertex[o * 9 + 7] = C.y; out.vertex[o * 9 + 8] = C.z;
      // vertex_attached_ holds the u32 label as a double; (int) of it is what
      // AddSTDescs / candidate_selector use (:158-160,362-364)
      out.label[o * 3 + 0] = (int)(double)__float_as_uint(A.w);
      out.label[o * 3 + 1] = (int)(double)__float_as_uint(B.w);
      out.label[o * 3 + 2] = (int)(double)__float_as_uint(C.w);
      out.frame[o] = frame_id;
      if (out.qrec) write_query_rec(out.qrec + o, cfg.scale * a, cfg.scale * b, cfg.scale * c, cfg.rough, frame_id);
      out.node_id[o * 3 + 0] = i; out.node_id[o * 3 + 1] = m; out.node_id[o * 3 + 2] = nn;
    }
#ifdef SGTD_EXP_PHASE
    __syncthreads();
    BPH(4);
#endif
  }
}

// LDS bytes the kernel carves for a batch whose largest frame has max_n keypoints
static inline size_t build_lds_bytes(int max_n, int K, int tpi, bool lds_dedup, int max_slots) {
  size_t max_t = (size_t)max_n * tpi, words = (max_t + 31) / 32;
  auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
  size_t b = (size_t)max_n * 16 + up((size_t)max_n * K * 2) + up((size_t)tpi * 2) +
             up(words * 4) + up((words + 1) * 4);
  if (lds_dedup) b += max_t * 8 + (size_t)max_slots * 4;
  return b;
}
