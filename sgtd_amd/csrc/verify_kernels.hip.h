// verify_kernels.hip.h — candidate_verify + triangle_solver
// (src/sgtd/src/STDesc.cpp:462-547 and :549-571), SURVEY §8f row 1.
//
// Two kernels, one workgroup per (query, candidate) each.  match_list_ of the candidate = the candidate's
// pair range written by block_write_kernel, in the reference's order.
//   hypotheses  (:467-468,481-487) every skip_len-th pair, use_size <= 50 of them: one thread
//               each solves the 3x3 Kabsch problem of its triangle pair (triangle_solver) — verify_solve_kernel
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
  double *hyp64;               // [nq * cand_num][SGTD_VERIFY_MAX_HYP][12] hypotheses (R row-major, t), pass 1 -> pass 2
  float *hyp32;                // [nq * cand_num][SGTD_VERIFY_MAX_HYP][24] the same in f32, every value twice
  u32 *bound;                  // [nq * cand_num][2] largest |rot entry| and |t|_1 of the candidate's hypotheses (float bits)
  double thr2;                 // smallest y with sqrt_rn(y) >= 3.0 (dis_threshold, :469)
  int exact_only;              // test hook (SGTD_VERIFY_EXACT=1): no f32 pre-test, every vertex A test in f64
  const u64 *keep;             // [nq] or NULL: bit c = verify candidate c of the query (sgtd_verify_masked: the candidates that
                               // survived a multi-GPU merge); the others score -1 like a rejected candidate
  // the matrix-core vote pass (verify_mfma.hip.h)
  u32 *words;                  // vote words: u32 [rows of 32 pairs][64], a candidate's rows start at (first pair / 32 + its index in the batch)
  uint4 *hypB;                 // [nq * cand_num][2][3][64] hypothesis features as the MFMAs' B operands
  float *tau;                  // [nq * cand_num][2][SGTD_VERIFY_MAX_HYP] |t|^2 (NaN: exact test only) and |t|_1 of the hypotheses
  const u32 *order;            // or NULL: the (query, candidate) indices in dispatch order (by candidate frame), n_blocks of them
  u32 n_blocks;
  u32 *inl_count;              // or NULL: [nq * cand_num] inlier pairs of every candidate (0 where there is no result) — what
                               // inlier_count_kernel would count from the flags (sgtd_search_frame: one launch less)
  const int *overflow;         // or NULL: the batch's overflow flags — set: the lists are not final (the batch will be re-run), nothing
                               // is read (sgtd_search_frame enqueues the verification behind the batch without a host round trip)
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

#define SGTD_VM_BLIMIT 2.5e4        // largest hypothesis feature the matrix pass accepts
// ---- the hypothesis side, written by verify_solve_kernel's thread h for hypothesis h (NaN tau: exact test only) -----------
// Layout = the B operands of the three MFMAs as the lanes read them: block (T, mi), T = h / 32, mi = 2 u + part; lane
// (h % 32) + 32 hh holds for k = 0..3 the feature of term 8 hh + 4 u + k, its high part (part 0) or low part (part 1), twice.
// (Measured against this layout in round 6, same box, back to back: the plain one — a block of high parts and a block of low
// parts on both sides, the products' high parts used by two of the three MFMAs, eight vector instructions and six registers
// less per tile — 12.26-12.30 ms per batch against 11.76-11.87: not adopted.)
__device__ __forceinline__ void vm_split(float x, _Float16 &hi, _Float16 &lo) {
  const float h = __uint_as_float(__float_as_uint(x) & 0xFFFFE000u);   // the top 11 significant bits
  hi = (_Float16)h;
  lo = (_Float16)(x - h);
}
__device__ inline void vm_write_hypothesis(const double out[12], bool solved, uint4 *hypB /* of the candidate */, float *tau /* of the candidate */, int h) {
  // b[4 i + j] = M_ij of  s (d^2 - 9) = sum_ij M_ij (w0, w1, w2, 1)_i s (v0, v1, v2, 1)_j + s (|v|^2 + |w|^2 - 9):
  // M = [ -2 R, -2 t ; 2 (R^T t)^T, |t|^2 ]
  double b[16];
  const double tt = (out[9] * out[9] + out[10] * out[10]) + out[11] * out[11];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) b[4 * i + j] = -2.0 * out[i * 3 + j];
    b[4 * i + 3] = -2.0 * out[9 + i];
  }
  for (int j = 0; j < 3; j++) b[12 + j] = 2.0 * (out[0 * 3 + j] * out[9] + out[1 * 3 + j] * out[10] + out[2 * 3 + j] * out[11]);
  b[15] = tt;
  bool ok = solved;
  double gmax = 0;      // |R^T R - I|_max
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double g = out[0 * 3 + i] * out[0 * 3 + j] + out[1 * 3 + i] * out[1 * 3 + j] + out[2 * 3 + i] * out[2 * 3 + j] - (i == j ? 1.0 : 0.0);
      gmax = fmax(gmax, fabs(g));
      if (!(g == g)) ok = false;
    }
  if (!(gmax <= 1e-6)) ok = false;
  for (int k = 0; k < 16; k++) if (!(fabs(b[k]) < SGTD_VM_BLIMIT)) ok = false;
  _Float16 hi[16], lo[16];
  for (int k = 0; k < 16; k++) {
    if (ok) vm_split((float)b[k], hi[k], lo[k]);
    else { hi[k] = (_Float16)0.0f; lo[k] = (_Float16)0.0f; }
  }
  tau[h] = ok ? (float)tt : __builtin_nanf("");
  {   // |t|_1, rounded up
    const double a1 = (fabs(out[9]) + fabs(out[10])) + fabs(out[11]);
    tau[SGTD_VERIFY_MAX_HYP + h] = ok ? (float)(a1 * 1.000001) : 0.0f;
  }
  // block (T, 0) and (T, 1): the high parts of the features of terms 8 hh + 0..3 and 8 hh + 4..7, each twice (against a product's
  // high and low part); block (T, 2): the low parts of terms 8 hh + 0..7, two to a word (against the products' high parts)
  const int T = h >> 5, c = h & 31;
  auto bits_of = [](_Float16 x) { unsigned short b16; __builtin_memcpy(&b16, &x, 2); return (u32)b16; };
  for (int hh = 0; hh < 2; hh++) {
    u32 wd[3][4];
    for (int k = 0; k < 4; k++) {
      wd[0][k] = bits_of(hi[8 * hh + k]) * 0x10001u;
      wd[1][k] = bits_of(hi[8 * hh + 4 + k]) * 0x10001u;
      wd[2][k] = bits_of(lo[8 * hh + 2 * k]) | (bits_of(lo[8 * hh + 2 * k + 1]) << 16);
    }
    for (int mi = 0; mi < 3; mi++) hypB[(size_t)(T * 3 + mi) * 64 + (size_t)(c + 32 * hh)] = make_uint4(wd[mi][0], wd[mi][1], wd[mi][2], wd[mi][3]);
  }
}

// keys of the dispatch order: the candidate's frame (candidates that do not exist last)
__global__ void verify_order_keys_kernel(const int *cand_frame, const int *n_cand, int cand_num, u32 n, u32 last, u32 *key, u32 *val) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int q = (int)(i / (u32)cand_num), c = (int)(i % (u32)cand_num);
  key[i] = c < n_cand[q] ? min((u32)cand_frame[i], last) : last;
  val[i] = i;
}

// ---- pass 1: the hypotheses (:467-468,481-487).  One 64-thread workgroup per (query, candidate): thread h
// solves the 3x3 Kabsch problem of pair h * skip_len.  Out: per hypothesis (R, t) in f64 (12 doubles) and
// the same rounded to f32 with every value TWICE (24 floats: the operands of the vote pass's packed f32
// chain as it reads them through the scalar data path), per (query, candidate) the bounds of the f32
// pre-test's error (largest |rot entry|, largest |t|_1, as float bit patterns; NaN: every pre-test undecided).
#define SGTD_HYP_F64 12
#define SGTD_HYP_F32 24
__global__ __launch_bounds__(SGTD_WAVE) void verify_solve_kernel(VerifyParams P) {
  const int tid = threadIdx.x;
  const int q = blockIdx.x / P.cand_num, c = blockIdx.x % P.cand_num;
  if (P.overflow && (P.overflow[0] | P.overflow[1])) return;
  if (c >= P.n_cand[q] || (P.keep && !((P.keep[q] >> c) & 1ull))) return;
  const long long *po = P.pair_off + (size_t)q * (P.cand_num + 1);
  const u32 base = P.q_pair_base[q] + (u32)po[c];
  const long long n = po[c + 1] - po[c];
  const int skip_len = (int)(n / 50) + 1;          // :467
  const int use_size = (int)(n / skip_len);        // :468
  const size_t qslot0 = (size_t)q * (size_t)P.q_stride;
  u32 rm_bits = P.exact_only ? 0x7FC00000u : 0u, tm_bits = 0u;
  double out[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < use_size) {
    double qv[9], ev[9], qc[3], ec[3];
    const u64 pr = P.pairs[base + (long long)tid * skip_len];
    const size_t qd = qslot0 + (size_t)(pr >> 32), g = (size_t)(pr & 0xFFFFFFFFull);
    for (int k = 0; k < 9; k++) { qv[k] = (double)P.q_vertex[qd * 9 + k]; ev[k] = (double)P.t_vertex[g * 9 + k]; }
    for (int k = 0; k < 3; k++) { qc[k] = P.q_center[qd * 3 + k]; ec[k] = P.t_center[g * 3 + k]; }
    solve_triangle_dev(qv, qc, ev, ec, out);
    double *h64 = P.hyp64 + ((size_t)blockIdx.x * SGTD_VERIFY_MAX_HYP + tid) * SGTD_HYP_F64;
    for (int k = 0; k < 12; k++) h64[k] = out[k];
    if (P.hyp32) {
      float *h32 = P.hyp32 + ((size_t)blockIdx.x * SGTD_VERIFY_MAX_HYP + tid) * SGTD_HYP_F32;
      for (int k = 0; k < 12; k++) { const float f = (float)out[k]; h32[2 * k] = f; h32[2 * k + 1] = f; }
    }
    // largest |rot entry| and |t|_1, rounded up, as float bit patterns (non-negative floats order like
    // their bits; a NaN beats everything)
    double rm = 0, tm = fabs(out[9]) + fabs(out[10]) + fabs(out[11]);
    for (int k = 0; k < 9; k++) rm = fmax(rm, fabs(out[k]));
    if (!(rm == rm)) rm = __builtin_nan("");
    rm_bits = max(rm_bits, __float_as_uint((float)(rm * 1.000001)));
    tm_bits = __float_as_uint((float)(tm * 1.000001));
  }
  if (P.hypB) vm_write_hypothesis(out, tid < use_size, P.hypB + (size_t)blockIdx.x * 6 * 64, P.tau + (size_t)blockIdx.x * 2 * SGTD_VERIFY_MAX_HYP, tid);
#pragma unroll
  for (int d = SGTD_WAVE / 2; d > 0; d >>= 1) {
    rm_bits = max(rm_bits, (u32)__shfl_xor((int)rm_bits, d));
    tm_bits = max(tm_bits, (u32)__shfl_xor((int)tm_bits, d));
  }
  if (tid == 0) { P.bound[2 * (size_t)blockIdx.x] = rm_bits; P.bound[2 * (size_t)blockIdx.x + 1] = tm_bits; }
}

// ---- pass 2: votes, best hypothesis, inliers.  The hypotheses are wave-uniform: they come through the
// scalar data path (written by the kernel before: the constant address space's promise holds) straight into
// the scalar operands of the packed f32 chain — no LDS broadcast reads (12 of them per step made the loop
// LDS-bound), no splat moves.
#ifndef SGTD_VERIFY_WAVES
#define SGTD_VERIFY_WAVES __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
#ifdef SGTD_EXP_VSTAT
// experiment build: how the (pair, hypothesis) tests fall out.  0 steps (128 pairs x one hypothesis), 1 steps where vertex A
// leaves some pair, 3 steps where it leaves eight or more, 7 valid (pair, hypothesis) combinations, 2 of them left by A,
// 5 of those in steps with eight or more, 6 combinations not far on all three vertices, 4 certain votes
__device__ unsigned long long g_vstat[8];
#define VSTAT(i, x) do { if (lane == 0) atomicAdd(&g_vstat[i], (unsigned long long)(x)); } while (0)
#else
#define VSTAT(i, x) do { } while (0)
#endif
typedef const __attribute__((address_space(4))) f32x2 *sgtd_const_f32x2;
typedef const __attribute__((address_space(4))) double *sgtd_const_f64;
// OR over the 64 lanes, the same value in every lane (the DPP steps of wave_incl_scan with | for +)
__device__ __forceinline__ u32 wave_or(u32 v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (u32)__builtin_amdgcn_readlane(x, SGTD_WAVE - 1);
}

// The (pair, hypothesis) combinations the f32 tests could not decide (per lane: bit masks), hypothesis by hypothesis
// for the whole wave: (R, t) in f64 through the scalar data path, the marked lanes test all three vertices with the exact
// squared threshold — the operations and their order are vertex_close's — and vote straight into the workgroup's LDS
// counters.  Behind the vote loop, and on vertices read again from memory: nothing of the loop's registers is live here.
__device__ __forceinline__ void verify_pending(const VerifyParams &P, sgtd_const_f64 hyp64, const float *const (&qp)[SGTD_VERIFY_PPT],
                                               const float *const (&ep)[SGTD_VERIFY_PPT], u64 (&pend)[SGTD_VERIFY_PPT],
                                               u64 (&passed)[SGTD_VERIFY_PPT], u32 *s_votes) {
  const u64 mine = pend[0] | pend[1];
  u64 any = ((u64)wave_or((u32)(mine >> 32)) << 32) | (u64)wave_or((u32)mine);
  while (any) {
    const int h = __builtin_ctzll(any);
    any &= any - 1ull;
    double Rt[12];
#pragma unroll
    for (int k = 0; k < 12; k++) Rt[k] = hyp64[h * SGTD_HYP_F64 + k];
#pragma unroll
    for (int u = 0; u < SGTD_VERIFY_PPT; u++) {
      if ((pend[u] >> h) & 1ull) {
        bool ok = true;
#pragma unroll 1
        for (int m = 0; m < 3; m++) {
          const double qa[3] = {(double)qp[u][3 * m], (double)qp[u][3 * m + 1], (double)qp[u][3 * m + 2]};
          const double ea[3] = {(double)ep[u][3 * m], (double)ep[u][3 * m + 1], (double)ep[u][3 * m + 2]};
          ok = ok && vertex_close(Rt, qa, ea, P.thr2);
        }
        if (ok) {
          passed[u] |= 1ull << h;
          atomicAdd(&s_votes[h], 1u);
        }
      }
    }
  }
}

__global__ __launch_bounds__(SGTD_VERIFY_THREADS) SGTD_VERIFY_WAVES void verify_kernel(VerifyParams P) {
  __shared__ u32 s_votes[SGTD_VERIFY_MAX_HYP];
  __shared__ u32 s_best, s_count;
  const int tid = threadIdx.x, lane = lane_id();
  const int q = blockIdx.x / P.cand_num, c = blockIdx.x % P.cand_num;
  double *score = P.score + (size_t)q * P.cand_num + c;
  if (P.overflow && (P.overflow[0] | P.overflow[1])) return;
  if (c >= P.n_cand[q] || (P.keep && !((P.keep[q] >> c) & 1ull))) { if (tid == 0) *score = -1.0; return; }
  const long long *po = P.pair_off + (size_t)q * (P.cand_num + 1);
  const u32 base = P.q_pair_base[q] + (u32)po[c];
  const u32 n = (u32)(po[c + 1] - po[c]);          // (a batch's pairs are indexed with 32 bits)
  const int skip_len = (int)(n / 50u) + 1;         // :467
  const int use_size = (int)(n / (u32)skip_len);   // :468
  const size_t qslot0 = (size_t)q * (size_t)P.q_stride;
  const sgtd_const_f32x2 hyp32 = (sgtd_const_f32x2)(unsigned long long)(P.hyp32 + (size_t)blockIdx.x * SGTD_VERIFY_MAX_HYP * SGTD_HYP_F32);
  const sgtd_const_f64 hyp64 = (sgtd_const_f64)(unsigned long long)(P.hyp64 + (size_t)blockIdx.x * SGTD_VERIFY_MAX_HYP * SGTD_HYP_F64);

  if (tid < SGTD_VERIFY_MAX_HYP) s_votes[tid] = 0;
  if (tid == 0) s_count = 0;
  __syncthreads();

  // ---- votes of every hypothesis (:488-505).  f32 pre-test of a vertex, both pairs of a thread
  // in the halves of packed operations (vertex B only where A passed, C only where B passed): p^ = fma chain of the f32-rounded (R, t) on the exact f32
  // vertices, d^ = fma(R0, v0, fma(R1, v1, fma(R2, v2, t - w))).  With u = 2^-24, |d^_i - d_i| <= 5 u (Rmax |v|_1 + |t|_1 + |w|_1)
  // (one rounding of R and t, one subtraction, three fused operations, every partial sum below that magnitude), the sum
  // of squares adds relative errors of a few u, so | ||d^|| - ||d|| | <= E = 16 u (Rmax |v|_1 + tmax + |w|_1 + thr) with room
  // to spare (the f64 evaluation of the reference is within 1e-15 relative of the exact value):
  //   d2^ < (thr - E)^2 (1 - 8u)  ==>  the reference's test passes
  //   d2^ > (thr + E)^2 (1 + 8u)  ==>  it fails;  anything else (NaN included) is decided in f64.
  u32 acc = 0;   // lane h of every wave: votes of hypothesis h seen by this wave
  constexpr int PPT = SGTD_VERIFY_PPT;
  static_assert(PPT == 2, "two pairs per thread: the halves of v_pk_*_f32");
  const float uf = 5.9604644775390625e-08f;
  const float rmaxf = __uint_as_float(P.bound[2 * (size_t)blockIdx.x]), tmaxf = __uint_as_float(P.bound[2 * (size_t)blockIdx.x + 1]);
  const float thrf = (float)(sqrt(P.thr2) * 1.000001);
  for (u32 j0 = 0; j0 < n; j0 += (u32)PPT * SGTD_VERIFY_THREADS) {
    bool valid[PPT];
    f32x2 v[3][3], w[3][3];        // [vertex A, B, C][x, y, z], the two pairs in the halves (f32 as stored)
    f32x2 lo2, hi2;                // the two squared f32 thresholds of each pair (from its largest vertex)
    u64 passed[PPT];                       // bit h: the pair votes for hypothesis h
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const u32 j = j0 + (u32)u * SGTD_VERIFY_THREADS + (u32)tid;
      valid[u] = j < n;
      const u64 pr = P.pairs[base + (valid[u] ? j : 0u)];
      const float *qp = P.q_vertex + (qslot0 + (size_t)(pr >> 32)) * 9, *ep = P.t_vertex + (size_t)(pr & 0xFFFFFFFFull) * 9;
      float vmax = 0.0f, wmax = 0.0f;
#pragma unroll
      for (int m = 0; m < 3; m++) {
        const float a0 = qp[3 * m], a1 = qp[3 * m + 1], a2 = qp[3 * m + 2];
        const float b0 = ep[3 * m], b1 = ep[3 * m + 1], b2 = ep[3 * m + 2];
        const float s1 = (fabsf(a0) + fabsf(a1)) + fabsf(a2), s2 = (fabsf(b0) + fabsf(b1)) + fabsf(b2);
        vmax = !(s1 <= vmax) ? s1 : vmax;                           // max that keeps a NaN
        wmax = !(s2 <= wmax) ? s2 : wmax;
        if (u == 0) { v[m][0].x = a0; v[m][1].x = a1; v[m][2].x = a2; w[m][0].x = b0; w[m][1].x = b1; w[m][2].x = b2; }
        else { v[m][0].y = a0; v[m][1].y = a1; v[m][2].y = a2; w[m][0].y = b0; w[m][1].y = b1; w[m][2].y = b2; }
      }
      const float E = 16.0f * uf * (rmaxf * vmax + tmaxf + wmax + thrf);
      const float lo = thrf * 0.999998f - E, hi = thrf + E;     // (thrf was rounded up by 1e-6: take it back for lo)
      const float l2 = lo > 0.0f ? lo * lo * (1.0f - 8.0f * uf) : 0.0f;  // NaN E: lo2 = 0 (never certainly in) ...
      const float h2 = hi * hi * (1.0f + 8.0f * uf);                     // ... and hi2 = NaN (never certainly out)
      if (u == 0) { lo2.x = l2; hi2.x = h2; } else { lo2.y = l2; hi2.y = h2; }
    }
    // Which of the wave's 2 x 64 pairs still vote for hypothesis h is kept as two 64-bit LANE MASKS in
    // scalar registers (not as per-lane booleans: those cost a v_cndmask / v_cmp pair at every use and
    // made this loop 140 VALU instructions per step; SQ counters, profiles/r03o_sq_verify.json).
    const u64 valid0 = __builtin_amdgcn_ballot_w64(valid[0]), valid1 = __builtin_amdgcn_ballot_w64(valid[1]);
#ifdef SGTD_EXP_VSTAT
    u64 any0 = 0, any1 = 0;
#endif
    // One step = one hypothesis against the wave's 2 x 64 pairs: the three vertices' squared f32 distances, their maximum
    // against the pair's two thresholds — certainly far (some vertex is: no vote), certainly close (all three are: a vote),
    // or undecided in f32: the pair is marked for the hypothesis and decided behind the loop, exactly as the reference
    // computes it (f64, all three vertices).  A pair with a NaN or an infinity anywhere has lo2 = 0 and hi2 = NaN / inf
    // (E above): never sure, never far.  The loop holds no f64 value.
    u64 pend[PPT] = {0ull, 0ull};          // bit h: (this pair, hypothesis h) is undecided in f32
    u32 pw[PPT][2] = {{0u, 0u}, {0u, 0u}}; // passed[u], low and high word
    auto steps = [&](int h_lo, int h_hi, u32 &p0, u32 &p1) {
      for (int h = h_lo; h < h_hi; h++) {
        f32x2 R[12];      // R00 R01 R02 R10 R11 R12 R20 R21 R22 t0 t1 t2, each in both halves (scalar registers)
#pragma unroll
        for (int k = 0; k < 12; k++) R[k] = hyp32[h * 12 + k];
        f32x2 d2[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
          // d = R v + (t - w): the table vertex goes into the chain's first addend (one operation per coordinate less
          // than (R v + t) - w; the scalar t and the vector w make one packed subtraction)
          const f32x2 dx = __builtin_elementwise_fma(R[0], v[m][0], __builtin_elementwise_fma(R[1], v[m][1], __builtin_elementwise_fma(R[2], v[m][2], R[9] - w[m][0])));
          const f32x2 dy = __builtin_elementwise_fma(R[3], v[m][0], __builtin_elementwise_fma(R[4], v[m][1], __builtin_elementwise_fma(R[5], v[m][2], R[10] - w[m][1])));
          const f32x2 dz = __builtin_elementwise_fma(R[6], v[m][0], __builtin_elementwise_fma(R[7], v[m][1], __builtin_elementwise_fma(R[8], v[m][2], R[11] - w[m][2])));
          d2[m] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        }
        const float mx = __builtin_fmaxf(__builtin_fmaxf(d2[0].x, d2[1].x), d2[2].x), my = __builtin_fmaxf(__builtin_fmaxf(d2[0].y, d2[1].y), d2[2].y);
        const u64 sure0 = __builtin_amdgcn_ballot_w64(mx < lo2.x), sure1 = __builtin_amdgcn_ballot_w64(my < lo2.y);
        const u64 far0 = __builtin_amdgcn_ballot_w64(mx > hi2.x), far1 = __builtin_amdgcn_ballot_w64(my > hi2.y);
        const u64 in0 = valid0 & ~far0, in1 = valid1 & ~far1;
        const u64 yes0 = in0 & sure0, yes1 = in1 & sure1, mark0 = in0 & ~sure0, mark1 = in1 & ~sure1;
        VSTAT(0, 1); VSTAT(7, __builtin_popcountll(valid0) + __builtin_popcountll(valid1)); VSTAT(4, __builtin_popcountll(yes0) + __builtin_popcountll(yes1));
#ifdef SGTD_EXP_VSTAT
        any0 |= in0; any1 |= in1;
        {   // vertex A alone: how many of the wave's pairs it leaves, and in how many steps it leaves any / eight or more
          const u64 a0 = valid0 & ~__builtin_amdgcn_ballot_w64(d2[0].x > hi2.x), a1 = valid1 & ~__builtin_amdgcn_ballot_w64(d2[0].y > hi2.y);
          const int na = __builtin_popcountll(a0) + __builtin_popcountll(a1);
          VSTAT(2, na); VSTAT(1, na > 0 ? 1 : 0); VSTAT(3, na >= 8 ? 1 : 0); VSTAT(5, na >= 8 ? na : 0); VSTAT(6, __builtin_popcountll(in0) + __builtin_popcountll(in1));
        }
#endif
        // the pair's bit of the hypothesis: the lane mask selects it (one v_cndmask and one v_or per pair)
        const u32 hb = 1u << (h & 31);
        u32 hv, t0, t1;
        asm volatile("v_mov_b32 %0, %1" : "=v"(hv) : "s"(hb));
        asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(t0) : "v"(hv), "s"(yes0));
        asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(t1) : "v"(hv), "s"(yes1));
        p0 |= t0; p1 |= t1;
        const u32 cnt = (u32)__builtin_popcountll(yes0) + (u32)__builtin_popcountll(yes1);
        if (lane == h) acc += cnt;
        if (mark0 | mark1) {     // rare
          if ((mark0 >> lane) & 1ull) pend[0] |= 1ull << h;
          if ((mark1 >> lane) & 1ull) pend[1] |= 1ull << h;
        }
      }
    };
    steps(0, use_size < 32 ? use_size : 32, pw[0][0], pw[1][0]);
    steps(32, use_size, pw[0][1], pw[1][1]);
#pragma unroll
    for (int u = 0; u < PPT; u++) passed[u] = ((u64)pw[u][1] << 32) | (u64)pw[u][0];
    // ---- the marked (pair, hypothesis) combinations, decided as the reference decides them (:488-505)
    if (__builtin_amdgcn_ballot_w64((pend[0] | pend[1]) != 0ull)) {      // rare: the pairs' vertices once more, from memory
      const float *qp[PPT], *ep[PPT];
#pragma unroll
      for (int u = 0; u < PPT; u++) {
        const u32 j = j0 + (u32)u * SGTD_VERIFY_THREADS + (u32)tid;
        const u64 pr = P.pairs[base + (j < n ? j : 0u)];
        qp[u] = P.q_vertex + (qslot0 + (size_t)(pr >> 32)) * 9;
        ep[u] = P.t_vertex + (size_t)(pr & 0xFFFFFFFFull) * 9;
      }
      verify_pending(P, hyp64, qp, ep, pend, passed, s_votes);
    }
    // what the inlier pass needs of this pair: no second walk over the vertices
#pragma unroll
    for (int u = 0; u < PPT; u++) {
      const u32 j = j0 + (u32)u * SGTD_VERIFY_THREADS + (u32)tid;
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
    for (u32 j = tid; j < n; j += SGTD_VERIFY_THREADS) P.inlier[base + j] = 0;
    return;
  }
  // ---- inliers of the best hypothesis (:516-539): the pairs that voted for it
  u32 mine = 0;
  for (u32 j = tid; j < n; j += SGTD_VERIFY_THREADS) {
    const bool in = (P.passed[base + j] >> best) & 1ull;
    P.inlier[base + j] = in ? 1 : 0;
    mine += in ? 1u : 0u;
  }
  mine = wave_sum(mine);
  if (lane == 0 && mine) atomicAdd(&s_count, mine);
  __syncthreads();
  if (tid == 0) *score = (double)s_count;           // :539
  if (tid < 12) P.pose[((size_t)q * P.cand_num + c) * 12 + tid] = P.hyp64[((size_t)blockIdx.x * SGTD_VERIFY_MAX_HYP + best) * SGTD_HYP_F64 + tid];
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

// ---- the same compaction by one workgroup per candidate (sgtd_search_frame: the one-frame-per-call path, where the
// 160 dependent rounds of the single workgroup above were a third of a millisecond): candidate k's inlier flags are
// counted by workgroup k, then every workgroup places its own list behind the counts of the candidates before it.
// `overflow` set: the batch's lists are not final, nothing is read.
#define SGTD_INLIER_CAND_THREADS 256
__global__ __launch_bounds__(SGTD_INLIER_CAND_THREADS) void inlier_count_kernel(const unsigned char *inlier, const long long *pair_off, const int *n_cand,
                                                                                const int *overflow, u32 *counts) {
  __shared__ u32 lds[SGTD_INLIER_CAND_THREADS / SGTD_WAVE + 1];
  const int k = blockIdx.x, tid = threadIdx.x;
  if (overflow[0] | overflow[1]) { if (tid == 0) counts[k] = 0; return; }
  u32 mine = 0;
  if (k < n_cand[0])
    for (long long i = pair_off[k] + tid; i < pair_off[k + 1]; i += SGTD_INLIER_CAND_THREADS) mine += inlier[i] ? 1u : 0u;
  u32 tot;
  (void)block_excl_scan(mine, lds, tot);
  if (tid == 0) counts[k] = tot;
}

__global__ __launch_bounds__(SGTD_INLIER_CAND_THREADS) void inlier_compact_kernel(const u64 *pairs, const unsigned char *inlier, const long long *pair_off,
                                                                                  const int *n_cand, const int *overflow, const u32 *counts,
                                                                                  int cand_num, u64 *out, long long *cand_off) {
  __shared__ u32 lds[SGTD_INLIER_CAND_THREADS / SGTD_WAVE + 1];
  __shared__ u32 s_base;
  const int k = blockIdx.x, tid = threadIdx.x;
  if (overflow[0] | overflow[1]) { if (tid == 0) { cand_off[k] = 0; if (k == 0) cand_off[cand_num] = 0; } return; }
  if (tid < SGTD_WAVE) {     // inlier pairs of the candidates before k (cand_num <= 64: one wave)
    const u32 c = tid < cand_num ? counts[tid] : 0u;
    const u32 before = wave_sum(tid < k ? c : 0u), all = wave_sum(c);
    if (tid == 0) {
      s_base = before;
      cand_off[k] = (long long)before;
      if (k == 0) cand_off[cand_num] = (long long)all;
    }
  }
  __syncthreads();
  if (k >= n_cand[0]) return;
  u32 carry = s_base;
  const long long lo = pair_off[k], hi = pair_off[k + 1];
  for (long long i0 = lo; i0 < hi; i0 += SGTD_INLIER_CAND_THREADS) {
    const long long i = i0 + tid;
    const u32 f = (i < hi && inlier[i]) ? 1u : 0u;
    u32 tot;
    const u32 ex = block_excl_scan(f, lds, tot);
    if (f) out[carry + ex] = pairs[i];
    carry += tot;
  }
}

// everything of a one-query batch the host needs after sgtd_search_frame's first (and usually only) wait, in one block:
//   u32 ctr[12] | i32 n_cand, u32 q_M, u32 pairs_total, u32 q_count | u64 q_P | i32 cand_frame[cn] | i32 cand_votes[cn] |
//   i64 pair_off[cn + 1] | f64 score[cn] | f64 pose[cn * 12] | i64 inl_off[cn + 1] | u64 totals[3] (behind frame_pack_bytes: the handle's running counts)
// `out` is page-locked host memory: the block is on the host when the stream has been waited for
__host__ __device__ __forceinline__ size_t frame_pack_bytes(int cn) {
  return 48 + 16 + 8 + (size_t)cn * 8 + (size_t)(cn + 1) * 8 + (size_t)cn * 8 + (size_t)cn * 96 + (size_t)(cn + 1) * 8;
}
__global__ __launch_bounds__(256) void pack_frame_kernel(const u32 *ctr, const int *n_cand, const u32 *q_M, const u32 *q_pair_base, const u32 *q_count,
                                                         const unsigned long long *q_P, const int *cand_frame, const int *cand_votes,
                                                         const long long *pair_off, const double *score, const double *pose, const long long *inl_off,
                                                         int cn, unsigned char *out, const unsigned long long *totals) {
  const int t = threadIdx.x;
  u32 *w = reinterpret_cast<u32 *>(out);
  if (t < 12) w[t] = ctr[t];
  if (t == 0) { w[12] = (u32)n_cand[0]; w[13] = q_M[0]; w[14] = q_pair_base[1]; w[15] = q_count[0]; *reinterpret_cast<unsigned long long *>(out + 64) = q_P[0]; }
  int *cf = reinterpret_cast<int *>(out + 72), *cv = cf + cn;
  long long *po = reinterpret_cast<long long *>(out + 72 + (size_t)cn * 8);
  double *sc = reinterpret_cast<double *>(po + cn + 1), *ps = sc + cn;
  long long *io = reinterpret_cast<long long *>(ps + (size_t)cn * 12);
  // (score / pose NULL: a call without verification — zeros)
  for (int k = t; k < cn; k += 256) { cf[k] = cand_frame[k]; cv[k] = cand_votes[k]; sc[k] = score ? score[k] : 0.0; }
  for (int k = t; k <= cn; k += 256) { po[k] = pair_off[k]; io[k] = inl_off[k]; }
  for (int k = t; k < cn * 12; k += 256) ps[k] = pose ? pose[k] : 0.0;
  if (t < 3) reinterpret_cast<unsigned long long *>(out + frame_pack_bytes(cn))[t] = totals ? totals[t] : 0ull;
}
