// verify_kernels.hip.h — candidate_verify + triangle_solver
// (src/sgtd/src/STDesc.cpp:462-547 and :549-571), SURVEY §8f row 1.
//
// One workgroup per (query, candidate).  match_list_ of the candidate = the candidate's
// pair range written by block_write_kernel, in the reference's order.
//   hypotheses  (:467-468,481-487) every skip_len-th pair, use_size <= 50 of them: one thread
//               each solves the 3x3 Kabsch problem of its triangle pair (triangle_solver)
//   votes       (:488-505) every thread owns two pairs of the current 512-pair tile and
//               tests them against all hypotheses (R, t broadcast from LDS); the per-hypothesis
//               counts are wave ballots accumulated in lane h.  A vertex is pre-tested in
//               packed f32 (both pairs in one v_pk_fma_f32 chain) against two squared
//               thresholds that make the f32 result conservative on both sides (bound
//               below); what falls between them is decided in f64 exactly as the reference
//               computes it.  Vertex A fails for nearly every (pair, hypothesis): B and C
//               are only looked at where it passes
//   best        (:507-514) first maximum; needs >= 4 votes (:515)
//   inliers     (:516-539) second pass with the best hypothesis: one flag byte per pair of
//               the list (sucess_match_vec = the flagged pairs in list order), score = count
//
// Arithmetic: f64, -ffp-contract=off.  Eigen::JacobiSVD (:559-561) is not available in this
// image; the 3x3 SVD is a one-sided (Hestenes) Jacobi SVD, the same operation order as the
// CPU restatement the parity tests check against — parity with Eigen's SVD in the reference
// binary is UNPINNED, parity with that restatement is exact.
#pragma once
#include "common.hip.h"

#define SGTD_VERIFY_THREADS 256
#define SGTD_VERIFY_MAX_HYP 64     // use_size <= 50 (:467-468)
#define SGTD_VERIFY_PPT 2          // pairs per thread per step of the vote pass: the halves of the packed f32 operations

struct VerifyParams {
  // candidate lists of the batch
  const u64 *pairs;            // q_idx << 32 | g
  const long long *pair_off;   // [nq][cand_num + 1]
  const u32 *q_pair_base;      // [nq]
  const int *n_cand;           // [nq]
  int cand_num;
  long long q_stride;          // descriptor slots per query
  // query descriptors (slot = q * q_stride + q_idx) and table entries (by g)
  const float *q_vertex;       // [.*9]
  const double *q_center;      // [.*3]
  const float *t_vertex;
  const double *t_center;
  // results
  double *score;               // [nq][cand_num]   verify_score (-1: fewer than 4 votes)
  double *pose;                // [nq][cand_num][12]  rot row-major (9) then t (3)
  unsigned char *inlier;       // [total pairs] flag per pair of every list
  u64 *passed;                 // [total pairs] scratch: bit h = the pair votes for hypothesis h of its candidate
  double thr2;                 // smallest y with sqrt_rn(y) >= 3.0 (dis_threshold, :469)
  int exact_only;              // test hook (SGTD_VERIFY_EXACT=1): no f32 pre-test, every vertex A test in f64
};

// One-sided (Hestenes) Jacobi SVD of a 3x3, H = U diag(s) V^T; columns of (near) zero
// singular values are completed by cross products so that U and V are orthogonal
__device__ inline void svd3_dev(const double H[3][3], double U[3][3], double V[3][3]) {
  double A[3][3], W[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { A[i][j] = H[i][j]; W[i][j] = (i == j) ? 1.0 : 0.0; U[i][j] = 0.0; V[i][j] = 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 3; k++) {
          alpha += A[k][p] * A[k][p];
          beta += A[k][q] * A[k][q];
          gamma += A[k][p] * A[k][q];
        }
        const double rel = fabs(gamma) / sqrt(alpha * beta + 1e-300);
        off = off < rel ? rel : off;               // std::max(off, rel)
        if (fabs(gamma) < 1e-300) continue;
        const double zeta = (beta - alpha) / (2 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
        const double c = 1 / sqrt(1 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) {
          const double ap = A[k][p], aq = A[k][q];
          A[k][p] = c * ap - s * aq;
          A[k][q] = s * ap + c * aq;
          const double wp = W[k][p], wq = W[k][q];
          W[k][p] = c * wp - s * wq;
          W[k][q] = s * wp + c * wq;
        }
      }
    if (off < 1e-15) break;
  }
  double s[3];
  int order[3] = {0, 1, 2};
  for (int j = 0; j < 3; j++) s[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  // std::sort(order, comp = s[a] > s[b]) on 3 elements = libstdc++ insertion sort
  for (int i = 1; i < 3; i++) {
    const int val = order[i];
    if (s[val] > s[order[0]]) {
      for (int j = i; j > 0; j--) order[j] = order[j - 1];
      order[0] = val;
    } else {
      int j = i;
      while (s[val] > s[order[j - 1]]) { order[j] = order[j - 1]; j--; }
      order[j] = val;
    }
  }
  const double smax = s[order[0]];
  int rank = 0;
  for (int jj = 0; jj < 3; jj++) {
    const int j = order[jj];
    for (int k = 0; k < 3; k++) V[k][jj] = W[k][j];
    if (s[j] > 1e-12 * (smax > 0 ? smax : 1)) {
      for (int k = 0; k < 3; k++) U[k][jj] = A[k][j] / s[j];
      rank = jj + 1;
    }
  }
  if (rank == 2) {
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  } else if (rank < 2) {
    for (int i = 0; i < 3; i++)
      for (int j = rank; j < 3; j++) U[i][j] = (i == j) ? 1 : 0;
  }
}

__device__ inline void mul3(const double a[3][3], const double b[3][3], double r[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

// triangle_solver (:549-571): out[0..8] = rot row-major, out[9..11] = t
__device__ inline void solve_triangle_dev(const double qv[9], const double qc[3], const double ev[9],
                                          const double ec[3], double out[12]) {
  double src[3][3], refT[3][3];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) {
      src[r][c] = qv[c * 3 + r] - qc[r];          // src.col(c) = vertex_c - center
      refT[c][r] = ev[c * 3 + r] - ec[r];         // ref.transpose()
    }
  double cov[3][3], U[3][3], V[3][3], UT[3][3], rot[3][3];
  mul3(src, refT, cov);                            // :558
  svd3_dev(cov, U, V);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) UT[i][j] = U[j][i];
  mul3(V, UT, rot);                                // :563
  const double det = rot[0][0] * (rot[1][1] * rot[2][2] - rot[1][2] * rot[2][1]) -
                     rot[0][1] * (rot[1][0] * rot[2][2] - rot[1][2] * rot[2][0]) +
                     rot[0][2] * (rot[1][0] * rot[2][1] - rot[1][1] * rot[2][0]);
  if (det < 0) {                                   // :564-568
    double K[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, -1}}, VK[3][3];
    mul3(V, K, VK);
    mul3(VK, UT, rot);
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out[r * 3 + c] = rot[r][c];
    out[9 + r] = -(rot[r][0] * qc[0] + rot[r][1] * qc[1] + rot[r][2] * qc[2]) + ec[r];   // :569
  }
}

// ||rot * v + t - w|| < 3.0 on squared values (exact: sq_threshold)
__device__ __forceinline__ bool vertex_close(const double *Rt, const double v[3], const double w[3], double thr2) {
  const double px = (Rt[0] * v[0] + Rt[1] * v[1] + Rt[2] * v[2]) + Rt[9];
  const double py = (Rt[3] * v[0] + Rt[4] * v[1] + Rt[5] * v[2]) + Rt[10];
  const double pz = (Rt[6] * v[0] + Rt[7] * v[1] + Rt[8] * v[2]) + Rt[11];
  const double dx = px - w[0], dy = py - w[1], dz = pz - w[2];
  return ((dx * dx + dy * dy) + dz * dz) < thr2;
}

// (four waves per SIMD: the solver's f64 temporaries must not set the register budget of the vote loop)
__global__ __launch_bounds__(SGTD_VERIFY_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void verify_kernel(VerifyParams P) {
  __shared__ double s_Rt[SGTD_VERIFY_MAX_HYP][12];
  __shared__ __attribute__((aligned(16))) float s_Rtf[SGTD_VERIFY_MAX_HYP][12];   // the same rounded to f32 (pre-test)
  __shared__ u32 s_votes[SGTD_VERIFY_MAX_HYP];
  __shared__ u32 s_best, s_count, s_rmax, s_tmax;
  const int tid = threadIdx.x, lane = lane_id();
  const int q = blockIdx.x / P.cand_num, c = blockIdx.x % P.cand_num;
  double *score = P.score + (size_t)q * P.cand_num + c;
  if (c >= P.n_cand[q]) { if (tid == 0) *score = -1.0; return; }
  const long long *po = P.pair_off + (size_t)q * (P.cand_num + 1);
  const u32 base = P.q_pair_base[q] + (u32)po[c];
  const long long n = po[c + 1] - po[c];
  const int skip_len = (int)(n / 50) + 1;          // :467
  const int use_size = (int)(n / skip_len);        // :468
  const size_t qslot0 = (size_t)q * (size_t)P.q_stride;

  auto load_pair = [&](long long j, double qv[9], double ev[9], size_t &qd, size_t &g) {
    const u64 pr = P.pairs[base + j];
    qd = qslot0 + (size_t)(pr >> 32);
    g = (size_t)(pr & 0xFFFFFFFFull);
    for (int k = 0; k < 9; k++) { qv[k] = (double)P.q_vertex[qd * 9 + k]; ev[k] = (double)P.t_vertex[g * 9 + k]; }
  };

  if (tid < SGTD_VERIFY_MAX_HYP) s_votes[tid] = 0;
  if (tid == 0) { s_count = 0; s_rmax = P.exact_only ? 0x7FC00000u : 0u; s_tmax = 0; }   // NaN bound: every pre-test undecided
  __syncthreads();
  if (tid < use_size) {
    double qv[9], ev[9], qc[3], ec[3], out[12];
    size_t qd, g;
    load_pair((long long)tid * skip_len, qv, ev, qd, g);
    for (int k = 0; k < 3; k++) { qc[k] = P.q_center[qd * 3 + k]; ec[k] = P.t_center[g * 3 + k]; }
    solve_triangle_dev(qv, qc, ev, ec, out);
    for (int k = 0; k < 12; k++) { s_Rt[tid][k] = out[k]; s_Rtf[tid][k] = (float)out[k]; }
    // largest |rot entry| and |t|_1 over the hypotheses, rounded up, as float bit patterns
    // (non-negative floats order like their bits; a NaN beats everything and makes every
    // f32 pre-test undecided)
    double rm = 0, tm = fabs(out[9]) + fabs(out[10]) + fabs(out[11]);
    for (int k = 0; k < 9; k++) rm = fmax(rm, fabs(out[k]));
    if (!(rm == rm)) rm = __builtin_nan("");
    atomicMax(&s_rmax, __float_as_uint((float)(rm * 1.000001)));
    atomicMax(&s_tmax, __float_as_uint((float)(tm * 1.000001)));
  }
  __syncthreads();

  // ---- votes of every hypothesis (:488-505).  f32 pre-test of a vertex, both pairs of a thread
  // in the halves of packed operations (vertex B only where A passed, C only where B passed): p^ = fma chain of the f32-rounded (R, t) on the exact f32
  // vertices.  With u = 2^-24, |p^_i - p_i| <= 4 u (Rmax |v|_1 + |t|_1) (one rounding of R and t,
  // three fused operations), the difference and the sum of squares add relative errors of a
  // few u, so | ||d^|| - ||d|| | <= E = 16 u (Rmax |v|_1 + tmax + thr) with room to spare (the
  // f64 evaluation of the reference is within 1e-15 relative of the exact value):
  //   d2^ < (thr - E)^2 (1 - 8u)  ==>  the reference's test passes
  //   d2^ > (thr + E)^2 (1 + 8u)  ==>  it fails;  anything else (NaN included) is decided in f64.
  u32 acc = 0;   // lane h of every wave: votes of hypothesis h seen by this wave
  constexpr int PPT = SGTD_VERIFY_PPT;
  static_assert(PPT == 2, "two pairs per thread: the halves of v_pk_*_f32");
  const float uf = 5.9604644775390625e-08f;
  const float rmaxf = __uint_as_float(s_rmax), tmaxf = __uint_as_float(s_tmax);
  const float thrf = (float)(sqrt(P.thr2) * 1.000001);
  for (long long j0 = 0; j0 < n; j0 += (long long)PPT * SGTD_VERIFY_THREADS) {
    bool valid[PPT];
    f32x2 v[3][3], w[3][3];        // [vertex A, B, C][x, y, z], the two pairs in the halves (f32 as stored)
    f32x2 lo2, hi2;                // the two squared f32 thresholds of each pair (from its largest vertex)
    u64 passed[PPT] = {0ull, 0ull};        // bit h: the pair votes for hypothesis h
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const long long j = j0 + (long long)u * SGTD_VERIFY_THREADS + tid;
      valid[u] = j < n;
      const u64 pr = P.pairs[base + (valid[u] ? j : 0)];
      const float *qp = P.q_vertex + (qslot0 + (size_t)(pr >> 32)) * 9, *ep = P.t_vertex + (size_t)(pr & 0xFFFFFFFFull) * 9;
      float vmax = 0.0f;
#pragma unroll
      for (int m = 0; m < 3; m++) {
        const float a0 = qp[3 * m], a1 = qp[3 * m + 1], a2 = qp[3 * m + 2];
        const float b0 = ep[3 * m], b1 = ep[3 * m + 1], b2 = ep[3 * m + 2];
        const float s1 = (fabsf(a0) + fabsf(a1)) + fabsf(a2);
        vmax = !(s1 <= vmax) ? s1 : vmax;                           // max that keeps a NaN
        if (u == 0) { v[m][0].x = a0; v[m][1].x = a1; v[m][2].x = a2; w[m][0].x = b0; w[m][1].x = b1; w[m][2].x = b2; }
        else { v[m][0].y = a0; v[m][1].y = a1; v[m][2].y = a2; w[m][0].y = b0; w[m][1].y = b1; w[m][2].y = b2; }
      }
      const float E = 16.0f * uf * (rmaxf * vmax + tmaxf + thrf);
      const float lo = thrf * 0.999998f - E, hi = thrf + E;     // (thrf was rounded up by 1e-6: take it back for lo)
      const float l2 = lo > 0.0f ? lo * lo * (1.0f - 8.0f * uf) : 0.0f;  // NaN E: lo2 = 0 (never certainly in) ...
      const float h2 = hi * hi * (1.0f + 8.0f * uf);                     // ... and hi2 = NaN (never certainly out)
      if (u == 0) { lo2.x = l2; hi2.x = h2; } else { lo2.y = l2; hi2.y = h2; }
    }
    for (int h = 0; h < use_size; h++) {
      const float4 r0 = reinterpret_cast<const float4 *>(s_Rtf[h])[0];   // R00 R01 R02 R10
      const float4 r1 = reinterpret_cast<const float4 *>(s_Rtf[h])[1];   // R11 R12 R20 R21
      const float4 r2 = reinterpret_cast<const float4 *>(s_Rtf[h])[2];   // R22 t0 t1 t2
      auto bc = [](float x) { f32x2 r = {x, x}; return r; };
      // vertex m of both pairs: certainly close (stays in), certainly far (leaves), or undecided in
      // f32 — then decided exactly, as the reference computes it
      auto vertex = [&](int m, bool (&in)[PPT]) {
        const f32x2 px = __builtin_elementwise_fma(bc(r0.x), v[m][0], __builtin_elementwise_fma(bc(r0.y), v[m][1], __builtin_elementwise_fma(bc(r0.z), v[m][2], bc(r2.y))));
        const f32x2 py = __builtin_elementwise_fma(bc(r0.w), v[m][0], __builtin_elementwise_fma(bc(r1.x), v[m][1], __builtin_elementwise_fma(bc(r1.y), v[m][2], bc(r2.z))));
        const f32x2 pz = __builtin_elementwise_fma(bc(r1.z), v[m][0], __builtin_elementwise_fma(bc(r1.w), v[m][1], __builtin_elementwise_fma(bc(r2.x), v[m][2], bc(r2.w))));
        const f32x2 dx = px - w[m][0], dy = py - w[m][1], dz = pz - w[m][2];
        const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        bool amb[PPT];
        amb[0] = in[0] && !(d2.x < lo2.x) && !(d2.x > hi2.x); in[0] = in[0] && d2.x < lo2.x;
        amb[1] = in[1] && !(d2.y < lo2.y) && !(d2.y > hi2.y); in[1] = in[1] && d2.y < lo2.y;
        if (__ballot(amb[0] || amb[1])) {     // rare
#pragma unroll
          for (int u = 0; u < PPT; u++)
            if (amb[u]) {
              const double qa[3] = {(double)(u ? v[m][0].y : v[m][0].x), (double)(u ? v[m][1].y : v[m][1].x), (double)(u ? v[m][2].y : v[m][2].x)};
              const double ea[3] = {(double)(u ? w[m][0].y : w[m][0].x), (double)(u ? w[m][1].y : w[m][1].x), (double)(u ? w[m][2].y : w[m][2].x)};
              in[u] = vertex_close(s_Rt[h], qa, ea, P.thr2);
            }
        }
      };
      bool in[PPT] = {valid[0], valid[1]};
      vertex(0, in);
      if (__ballot(in[0] || in[1])) {       // wrong hypotheses fail at vertex A for the whole wave
        vertex(1, in);
        if (__ballot(in[0] || in[1])) vertex(2, in);
        if (in[0]) passed[0] |= 1ull << h;
        if (in[1]) passed[1] |= 1ull << h;
        const u32 cnt = (u32)__popcll(__ballot(in[0])) + (u32)__popcll(__ballot(in[1]));
        if (lane == h) acc += cnt;
      }
    }
    // what the inlier pass needs of this pair: no second walk over the vertices
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const long long j = j0 + (long long)u * SGTD_VERIFY_THREADS + tid;
      if (j < n) P.passed[base + j] = passed[u];
    }
  }
  if (lane < use_size && acc) atomicAdd(&s_votes[lane], acc);
  __syncthreads();
  if (tid < SGTD_WAVE) {   // first maximum (:507-514): most votes, then lowest index
    u64 key = (tid < use_size) ? (((u64)s_votes[tid] << 8) | (u64)(63 - tid)) : 0ull;
#pragma unroll
    for (int d = SGTD_WAVE / 2; d > 0; d >>= 1) {
      const u64 o = __shfl_xor(key, d);
      key = o > key ? o : key;
    }
    if (tid == 0) s_best = ((u32)(key >> 8) >= 4u) ? (u32)(63 - (int)(key & 0xFF)) : 0xFFFFFFFFu;   // :515
  }
  __syncthreads();
  const u32 best = s_best;
  if (best == 0xFFFFFFFFu) {
    if (tid == 0) *score = -1.0;                    // :541
    for (long long j = tid; j < n; j += SGTD_VERIFY_THREADS) P.inlier[base + j] = 0;
    return;
  }
  // ---- inliers of the best hypothesis (:516-539): the pairs that voted for it
  const double *Rt = s_Rt[best];
  u32 mine = 0;
  for (long long j = tid; j < n; j += SGTD_VERIFY_THREADS) {
    const bool in = (P.passed[base + j] >> best) & 1ull;
    P.inlier[base + j] = in ? 1 : 0;
    mine += in ? 1u : 0u;
  }
  mine = wave_sum(mine);
  if (lane == 0 && mine) atomicAdd(&s_count, mine);
  __syncthreads();
  if (tid == 0) *score = (double)s_count;           // :539
  if (tid < 12) P.pose[((size_t)q * P.cand_num + c) * 12 + tid] = Rt[tid];
}

// SearchLoop's choice among the verified candidates (:105-146): the first candidate with
// the strictly largest score, accepted if score > icp_threshold
__global__ void search_loop_kernel(const double *score, const int *cand_frame, const int *n_cand, int cand_num,
                                   int nq, double icp_threshold, int *best_cand, int *best_frame, double *best_score) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  double bs = 0;
  int bc = -1;
  for (int c = 0; c < n_cand[q]; c++) {
    const double s = score[(size_t)q * cand_num + c];
    if (s > bs) { bs = s; bc = c; }
  }
  if (bs > icp_threshold) {
    best_cand[q] = bc; best_frame[q] = cand_frame[(size_t)q * cand_num + bc]; best_score[q] = bs;
  } else {
    best_cand[q] = -1; best_frame[q] = -1; best_score[q] = 0;   // loop_result = (-1, 0) (:144)
  }
}


// The inlier pairs (sucess_match_vec, STDesc.cpp:516-539) of every candidate of ONE query, compacted in
// match-list order: out[j] = the j-th pair whose inlier flag is set, cand_off[k] = inlier pairs before
// candidate k's list (cand_off[cand_num] = all of them).  One workgroup walks the query's pairs.
#define SGTD_INLIER_THREADS 1024
__global__ __launch_bounds__(SGTD_INLIER_THREADS) void inlier_pairs_kernel(const u64 *pairs, const unsigned char *inlier,
                                                                           const long long *pair_off /* of the query, [cand_num + 1] */,
                                                                           int cand_num, u64 *out, long long *cand_off) {
  __shared__ u32 lds[SGTD_INLIER_THREADS / SGTD_WAVE + 1];
  __shared__ u32 s_ex[SGTD_INLIER_THREADS];
  const long long total = pair_off[cand_num];
  const int tid = threadIdx.x;
  const long long bound = tid <= cand_num ? pair_off[tid] : -1;      // thread k watches candidate k's first pair
  u32 carry = 0;
  for (long long i0 = 0; i0 < total; i0 += SGTD_INLIER_THREADS) {
    const long long i = i0 + tid;
    const u32 f = (i < total && inlier[i]) ? 1u : 0u;
    u32 tot;
    const u32 ex = block_excl_scan(f, lds, tot);
    if (f) out[carry + ex] = pairs[i];
    s_ex[tid] = ex;
    __syncthreads();
    if (bound >= i0 && bound < i0 + SGTD_INLIER_THREADS && bound < total) cand_off[tid] = (long long)(carry + s_ex[bound - i0]);
    __syncthreads();
    carry += tot;
  }
  if (bound >= 0 && bound >= total) cand_off[tid] = (long long)carry;    // lists that start at the end (empty ones, the closing offset)
}
