// verify_mfma.hip.h — the vote pass of candidate_verify (src/sgtd/src/STDesc.cpp:488-505) on the matrix cores.
//
// Every (pair, hypothesis) of a candidate asks: are the three vertices of the query triangle, moved by the hypothesis,
// within 3 m of the table triangle's (:492-503)?  On the benchmark's maps nine of ten such tests END IN A VOTE (the fifty
// candidates of a query frame are the frames around the place, every one of them a rigid view of the same landmarks:
// profiles/r06a_verify_profile.json), so no early exit saves anything: all three squared distances of all
// n x use_size combinations are needed, 1.6e10 of them per batch — a dense contraction.  Expanded,
//
//   |R v + t - w|^2 - 9  =  sum_ij (-2 R_ij)(w_i v_j) + sum_j (2 (R^T t)_j) v_j + sum_i (-2 t_i) w_i + (|v|^2 + |w|^2 - 9) + |t|^2
//
// (|R v|^2 = |v|^2: the solve kernel measures |R^T R - I|_max and leaves a hypothesis beyond 1e-6 to the exact test), a
// bilinear form of sixteen PAIR features ((w0, w1, w2, 1)_i x s (v0, v1, v2, 1)_j per vertex; the pair's own term
// kap = s (|v|^2 + |w|^2 - 9) starts the accumulators instead) and sixteen HYPOTHESIS features (M = [-2 R, -2 t; 2 (R^T t)^T,
// |t|^2], written by the solve kernel as the MFMAs' B operands: vm_write_hypothesis).  One v_mfma_f32_32x32x16_f16 multiplies
// 32 pairs by 32 hypotheses over 16 products; every feature enters as TWO f16 values (x = hi + lo, hi = the top 11
// significant bits, lo = the next 11) and three of the four partial products are summed — hi x hi, lo x hi, hi x lo; lo x lo
// is below 2^-20 of the product and belongs to the error — so a product carries 3 x 2^-20 of relative error, not 2^-11.
// Three MFMAs per (vertex, 32 hypotheses) give the 32 x 32 values in 16 registers per lane — lane = hypothesis, register =
// pair — which the vector unit only has to take the maximum of over the three vertices and look at the sign of.
//
// The result of the matrix pass is only ever used where it is CERTAIN.  With the pair's features scaled by s (a power of
// two, the same for the 32 pairs of a tile) the lane has zhat = s (d^2 - 9) up to an error below e for the worst of the
// three vertices, e = s EPS B_h + 0.12 for the lane's hypothesis (B_h below), hence with z = zhat + e
//     z < 0     ==>  every vertex has d^2 < 9: the reference's three tests pass, a vote
//     z > 2 e   ==>  some vertex has d^2 > 9: no vote
// and whatever lies between (or is not a number) is queued and decided exactly as the reference computes it — f64,
// vertex_close of verify_kernels.hip.h — so that every decision equals the reference's (0.2 % of the combinations on the
// benchmark's batch, profiles/r06v_verify_profile.json).
//
// Error of zhat.  Let V = max_m |v_m|_1, W = max_m |w_m|_1 over the tile's pairs and vertices, T = |t|_1 of the hypothesis
// and rho = max(1, max |R_ij|) over the candidate's hypotheses.  The absolute values of the terms of the expansion sum to at
// most B_h = (rho V + T + W)^2 + 16.  Sources of error, relative to B_h: the two-part f16 representation (each factor's low
// part truncated: 2^-20 per factor; the product lo x lo left out: 2^-20: 2.9e-6 together), the f32 roundings of the pair
// features (3 x 2^-24), the f32 accumulation of 48 products and kap per value (at most one rounding of at most 2^-23 B_h
// each, whatever the order the matrix pipe sums in: 5.8e-6), the orthogonality defect (1e-6).  Together below 1.0e-5 B_h;
// SGTD_VM_EPS = 1.3e-5 is used.  s is the largest power of two <= 2.9 / (EPS B_max), B_max with the candidate's largest |t|_1:
// the scaled error s EPS B_h stays below 2.9; the f16 subnormal grid (2^-24 absolute on a part, times a partner below 2.5e4:
// 0.0015 per product, 48 products: the parts that can be subnormal are low parts, whose partners are far smaller — 0.04 is a
// generous total) and the roundings of kap and of the final additions are the 0.12.  No feature can overflow f16:
// s |w_i v_j| <= s V W <= 2.9 / (4 EPS) = 55 800 < 65 504 (f16's largest); hypothesis features are checked by the solve
// kernel (|b| < 2.5e4, else the hypothesis is left to the exact test: tau = NaN), a pair with a coordinate that is not a
// number or beyond 1e6 is left to it as well.
#pragma once
#include "verify_kernels.hip.h"

typedef _Float16 sgtd_h8 __attribute__((ext_vector_type(8)));
typedef float sgtd_f32x16 __attribute__((ext_vector_type(16)));

#define SGTD_VM_THREADS 256
#define SGTD_VM_QCAP 1024           // per wave: (pair, hypothesis) combinations waiting for the exact test (>= 64 x 16: one tile's)
#define SGTD_VM_EPS 1.3e-5f
#ifndef SGTD_VM_WAVES
#define SGTD_VM_WAVES 2
#endif
// SGTD_VM_PIPE (experiment switch): 1 / 2 = a software pipeline across tiles — the next tile prepared between the current tile's
// MFMAs (one hypothesis tile's / both tiles', the order forced with sched_group_barrier).  Measured and not the default: 13.8 / 13.1
// against 12.7 ms per batch on the same box — the kernel is bound by the NUMBER of vector instructions (4 cycles each on a SIMD
// whichever wave they come from), not by where the MFMAs sit among them.
#ifndef SGTD_VM_PIPE
#define SGTD_VM_PIPE 0
#endif
#ifndef SGTD_VM_VALU_PER_MFMA
#define SGTD_VM_VALU_PER_MFMA 18
#endif

#ifdef SGTD_EXP_VSTAT
// experiment build: 0 (pair tile, hypothesis tile) steps, 1 pair tiles with something queued, 2 combinations queued for the
// exact test, 3 of them votes, 4 drains of a wave's queue, 5 certain votes
__device__ unsigned long long g_vmstat[8];
#define VMSTAT(i, x) do { if (lane == 0) atomicAdd(&g_vmstat[i], (unsigned long long)(x)); } while (0)
#define VMSTAT1(i, x) atomicAdd(&g_vmstat[i], (unsigned long long)(x))
#else
#define VMSTAT(i, x) do { } while (0)
#define VMSTAT1(i, x) do { } while (0)
#endif

// pair row of register g in the lanes' half hh of a 32 x 32 result, and back
__device__ __forceinline__ u32 vm_row(u32 g, u32 hh) { return 8u * (g >> 2) + 4u * hh + (g & 3u); }
// bits g = 0..15 of the 32-row mask x that belong to the half hh
__device__ __forceinline__ u32 vm_rows16(u32 x, u32 hh) {
  x >>= 4u * hh;
  return (x & 0xFu) | ((x >> 4) & 0xF0u) | ((x >> 8) & 0xF00u) | ((x >> 12) & 0xF000u);
}
// The exact test of what the matrix pass left open, 64 combinations at a time (every lane its own pair and hypothesis; queue
// entry = pair index in the list << 8 | hypothesis): the reference's computation (vertex_close, f64).  A vote sets the
// combination's bit in the candidate's vote words and counts.
__device__ __forceinline__ void vm_drain(const VerifyParams &P, u32 bid, const u64 *queue, u32 qn, u32 base, size_t qslot0, u32 *words, u32 *s_votes) {
  const int lane = lane_id();
  // the words of earlier tiles have reached L2 before their bits are set there, the queue's entries are written before other
  // lanes read them (workgroup scope: a wait for the wave's own stores — an agent-scope release would write the L2 back)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (u32 i0 = 0; i0 < qn; i0 += SGTD_WAVE) {
    const u32 i = i0 + (u32)lane;
    if (i < qn) {
      const u64 en = queue[i];
      const u32 j = (u32)(en >> 8), h = (u32)(en & 0xFFu);
      const u64 pr = P.pairs[base + j];
      const float *qp = P.q_vertex + (qslot0 + (size_t)(pr >> 32)) * 9, *ep = P.t_vertex + (size_t)(pr & 0xFFFFFFFFull) * 9;
      const double *Rt = P.hyp64 + ((size_t)bid * SGTD_VERIFY_MAX_HYP + h) * SGTD_HYP_F64;
      double R[12];
#pragma unroll
      for (int k = 0; k < 12; k++) R[k] = Rt[k];
      bool ok = true;
#pragma unroll 1
      for (int m = 0; m < 3; m++) {
        const double qa[3] = {(double)qp[3 * m], (double)qp[3 * m + 1], (double)qp[3 * m + 2]};
        const double ea[3] = {(double)ep[3 * m], (double)ep[3 * m + 1], (double)ep[3 * m + 2]};
        ok = ok && vertex_close(R, qa, ea, P.thr2);
      }
      VMSTAT1(2, 1);
      if (ok) {
        const u32 row = j & 31u, h2 = (row >> 2) & 1u, g = ((row >> 3) << 2) | (row & 3u);
        atomicOr(&words[(size_t)(j >> 5) * 64 + (h & 31u) + 32u * h2], 1u << (g + 16u * (h >> 5)));
        atomicAdd(&s_votes[h], 1u);
        VMSTAT1(3, 1);
      }
    }
  }
}

// ---- pass 2 on the matrix cores: votes, best hypothesis, inliers.  One workgroup per (query, candidate); its four
// waves take the candidate's 32-pair tiles in turn.  Vote words: u32 [tile][64]; the word of lane (h % 32) + 32 hh holds
// in bit g + 16 (h / 32) the vote of pair 32 tile + vm_row(g, hh) for hypothesis h.
// (at least two waves per SIMD: within 256 registers the compiler keeps the MFMA results in VGPRs — with AGPR results every
// register of them costs a v_accvgpr_read before the vector unit can look at it)
// NW waves per workgroup: four for a batch (hundreds of candidates per CU), eight for a frame's fifty candidates (fifty workgroups on
// 256 CUs: twice the waves halve the tiles each walks)
template <int NW>
__global__ __launch_bounds__(NW * SGTD_WAVE) __attribute__((amdgpu_waves_per_eu(SGTD_VM_WAVES, 4))) void verify_mfma_kernel(VerifyParams P) {
  constexpr u32 THREADS = NW * SGTD_WAVE;
  __shared__ u32 s_votes[SGTD_VERIFY_MAX_HYP];
  __shared__ u32 s_best, s_count;
  __shared__ u64 s_queue[NW][SGTD_VM_QCAP];
  __shared__ __attribute__((aligned(16))) float s_kap[NW][2][96];      // per wave: kap of two tiles x three vertices x 32 pairs
  const int tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (the wave's number in a scalar register)
  const u32 r = (u32)lane & 31u, hh = (u32)lane >> 5;
  if (P.overflow && (P.overflow[0] | P.overflow[1])) return;
  // which (query, candidate) this workgroup takes.  With a dispatch order (a batch of many candidates, sorted by candidate
  // frame): workgroup b runs on XCD b % 8, and every XCD walks its own eighth of the order front to back — candidates on the
  // same map frame gather the same 260 KB of table vertices, and now do so next to each other in one L2 (and in the
  // Infinity Cache) instead of fifty workgroups apart
  u32 bid = blockIdx.x;
  if (P.order) {
    const u32 per = gridDim.x >> 3, slot = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (slot >= P.n_blocks) return;
    bid = P.order[slot];
  }
  const int q = (int)(bid / (u32)P.cand_num), c = (int)(bid % (u32)P.cand_num);
  double *score = P.score + (size_t)q * P.cand_num + c;
  double *pose = P.pose + ((size_t)q * P.cand_num + c) * 12;      // (zero where there is no result: no memset before the kernel)
  if (c >= P.n_cand[q] || (P.keep && !((P.keep[q] >> c) & 1ull))) {
    if (tid == 0) { *score = -1.0; if (P.inl_count) P.inl_count[bid] = 0u; }
    if (tid < 12) pose[tid] = 0.0;
    return;
  }
  const long long *po = P.pair_off + (size_t)q * (P.cand_num + 1);
  const u32 base = P.q_pair_base[q] + (u32)po[c];
  const u32 n = (u32)(po[c + 1] - po[c]);          // (a batch's pairs are indexed with 32 bits)
  const int skip_len = (int)(n / 50u) + 1;         // :467
  const int use_size = (int)(n / (u32)skip_len);   // :468
  const size_t qslot0 = (size_t)q * (size_t)P.q_stride;
  // the candidate's vote words: rows of 64 words, one row per 32 pairs.  Candidate k of the batch starts at row
  // base / 32 + k: floor(a / 32) + ceil(n / 32) <= floor((a + n) / 32) + 1, so consecutive candidates never share a row
  u32 *words = P.words + (((size_t)base >> 5) + (size_t)bid) * 64;
  u64 *queue = s_queue[wave];

  if (tid < SGTD_VERIFY_MAX_HYP) s_votes[tid] = 0;
  if (tid == 0) s_count = 0;
  __syncthreads();

  // ---- the hypothesis side of the four MFMAs, for both hypothesis tiles (registers for the whole candidate)
  const float rmaxf = __uint_as_float(P.bound[2 * (size_t)bid]), tmaxf = __uint_as_float(P.bound[2 * (size_t)bid + 1]);
  const bool cand_exact = P.exact_only || !(rmaxf < 1e3f) || !(tmaxf < 1e5f);     // (NaN bounds: everything exact)
  const float rho = fmaxf(1.0f, rmaxf);
  const int n_ht = use_size > 32 ? 2 : 1;
  sgtd_h8 Bop[2][3];
  float t1[2];              // |t|_1 of this lane's two hypotheses
  bool hyp_ok[2], hyp_exact[2];
  {
    const uint4 *hb = P.hypB + (size_t)bid * 6 * 64;
#pragma unroll
    for (int T = 0; T < 2; T++) {
      const int h = T * 32 + (int)r;
      hyp_ok[T] = h < use_size;
      const float tv = hyp_ok[T] ? P.tau[(size_t)bid * 2 * SGTD_VERIFY_MAX_HYP + h] : 0.0f;
      hyp_exact[T] = hyp_ok[T] && (cand_exact || !(tv == tv));
      t1[T] = hyp_exact[T] || !hyp_ok[T] ? 0.0f : P.tau[(size_t)bid * 2 * SGTD_VERIFY_MAX_HYP + SGTD_VERIFY_MAX_HYP + h];
#pragma unroll
      for (int mi = 0; mi < 3; mi++) {
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        if (T < n_ht) x = hb[(size_t)(T * 3 + mi) * 64 + lane];
        __builtin_memcpy(&Bop[T][mi], &x, 16);
      }
    }
  }

  u32 cnt[2] = {0u, 0u};     // certain votes of this lane's hypotheses (one per tile T) among the rows of its half
  u32 qn = 0;                // entries in the wave's queue
  float *kap_w = s_kap[wave][0];
  constexpr u32 STEP = NW;

  // the tile's 32 pairs: lanes r and r + 32 both hold pair r (they feed different products of it).  Loaded one tile ahead
  // (the vertices) and two ahead (the pair words): the gather's two dependent round trips never stand in the loop
  struct Tile { float v[3][3], w[3][3]; };
  auto pair_of = [&](u32 t) -> u64 { const u32 jj = t * 32u + r; return P.pairs[base + (jj < n ? jj : 0u)]; };
  auto vertices_of = [&](u64 pr, Tile &d) {
    const float *qp = P.q_vertex + (qslot0 + (size_t)(pr >> 32)) * 9, *ep = P.t_vertex + (size_t)(pr & 0xFFFFFFFFull) * 9;
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
      for (int k = 0; k < 3; k++) { d.v[m][k] = qp[3 * m + k]; d.w[m][k] = ep[3 * m + k]; }
  };
  // (past the wave's last tile the loads go to that tile again: no branch in the loop, so the compiler counts the loads in flight
  // instead of waiting for all of them)
  const u32 n_tiles = (n + 31u) >> 5;
  const u32 last_tile = n_tiles > (u32)wave ? (u32)wave + ((n_tiles - 1u - (u32)wave) / STEP) * STEP : (u32)wave;
  // what a tile's pairs become before the matrix pipe sees them: the tile's scale and masks, and per vertex the two-part products
  // (the MFMAs' A operands) and kap (through LDS: a lane needs it for its sixteen ROWS)
  struct Prep { float Umax, s, se; u32 off16, wild16; sgtd_h8 Aop[3][3]; };
  auto prepare = [&](const Tile &t, u32 tile, float *kap_buf, Prep &p) {
    const bool valid = tile * 32u + r < n;
    const float (&v)[3][3] = t.v, (&w)[3][3] = t.w;
    // V = max_m |v_m|_1, W = max_m |w_m|_1 (one v_add with |.| on both inputs and one more per vertex); the sum of all of them
    // catches what a maximum drops: a NaN
    float s1[3], s2[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
      asm("v_add_f32 %0, |%1|, |%2|" : "=v"(s1[m]) : "v"(v[m][0]), "v"(v[m][1]));
      asm("v_add_f32 %0, %1, |%2|" : "=v"(s1[m]) : "v"(s1[m]), "v"(v[m][2]));
      asm("v_add_f32 %0, |%1|, |%2|" : "=v"(s2[m]) : "v"(w[m][0]), "v"(w[m][1]));
      asm("v_add_f32 %0, %1, |%2|" : "=v"(s2[m]) : "v"(s2[m]), "v"(w[m][2]));
    }
    const float V = __builtin_fmaxf(__builtin_fmaxf(s1[0], s1[1]), s1[2]), W = __builtin_fmaxf(__builtin_fmaxf(s2[0], s2[1]), s2[2]);
    const float all = ((s1[0] + s1[1]) + s1[2]) + ((s2[0] + s2[1]) + s2[2]);
    // a pair the matrix pass cannot take (not a number, or beyond every sensible coordinate): exact test for all its hypotheses.
    // Its features may be anything: a row of the A operand only ever reaches its own row of the result, which is masked out
    const bool wild = valid && !(all < 1e6f);
    // rho V + W: with a hypothesis's |t|_1 the bound B of the error analysis (masked, not selected by a branch: the block must stay ONE
    // scheduling region for the interleaving below)
    const float U = __uint_as_float(__float_as_uint(__builtin_fmaf(rho, V, W)) & ((valid && !wild) ? 0xFFFFFFFFu : 0u));
    p.Umax = __uint_as_float(wave_max_u32(__float_as_uint(U)));               // (non-negative floats order like their bit patterns)
    // s = a power of two <= 3 / (EPS B) for the candidate's largest |t|_1 (scaling by it is exact; rcp's last bits are covered by
    // the 2.9): no feature exceeds s V W <= 3 / (4 EPS) = 57 700
    const float bmax = (p.Umax + tmaxf) * (p.Umax + tmaxf) + 16.0f;
    const float s = __uint_as_float(__float_as_uint(2.9f * __builtin_amdgcn_rcpf(SGTD_VM_EPS * 1.0001f * bmax)) & 0x7F800000u);
    p.s = s;
    p.se = s * (SGTD_VM_EPS * 1.0001f);
    const u32 off_rows = (u32)__builtin_amdgcn_ballot_w64(!valid || wild);           // rows without a matrix result (low 32 bits: lanes 0..31)
    const u32 wild_rows = (u32)__builtin_amdgcn_ballot_w64(wild);
    p.off16 = vm_rows16(off_rows, hh); p.wild16 = vm_rows16(wild_rows, hh);
    // the products (w0, w1, w2, 1)_i x s (v0, v1, v2, 1)_j, rows i = 2 hh, 2 hh + 1 in this lane, each as a (high part, low part)
    // pair of f16; kap = s (|v|^2 + |w|^2 - 9) enters through the accumulators' initial value
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const f32x2 sv01 = (f32x2){v[m][0], v[m][1]} * (f32x2){s, s}, sv2s = (f32x2){v[m][2], 1.0f} * (f32x2){s, s};
      const float kap = __builtin_fmaf(v[m][0], sv01.x, __builtin_fmaf(v[m][1], sv01.y, __builtin_fmaf(v[m][2], sv2s.x,
                        s * __builtin_fmaf(w[m][0], w[m][0], __builtin_fmaf(w[m][1], w[m][1], __builtin_fmaf(w[m][2], w[m][2], -9.0f))))));
      kap_buf[m * 32 + (int)r] = kap;      // (both lanes of the pair: the same value to the same word)
      const float wa = hh ? w[m][2] : w[m][0], wb = hh ? 1.0f : w[m][1];
      const f32x2 x[4] = {(f32x2){wa, wa} * sv01, (f32x2){wa, wa} * sv2s, (f32x2){wb, wb} * sv01, (f32x2){wb, wb} * sv2s};
      u32 wd[8], hw[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const f32x2 hi = (f32x2){__uint_as_float(__float_as_uint(x[k].x) & 0xFFFFE000u), __uint_as_float(__float_as_uint(x[k].y) & 0xFFFFE000u)};
        const f32x2 lo = x[k] - hi;
        auto p0 = __builtin_amdgcn_cvt_pkrtz(hi.x, lo.x);        // (hi, lo) of a product: against the high part of its hypothesis feature, twice
        auto p1 = __builtin_amdgcn_cvt_pkrtz(hi.y, lo.y);
        auto p2 = __builtin_amdgcn_cvt_pkrtz(hi.x, hi.y);        // the high parts of two products: against the low parts of their features
        __builtin_memcpy(&wd[2 * k], &p0, 4);
        __builtin_memcpy(&wd[2 * k + 1], &p1, 4);
        __builtin_memcpy(&hw[k], &p2, 4);
      }
      __builtin_memcpy(&p.Aop[m][0], &wd[0], 16);
      __builtin_memcpy(&p.Aop[m][1], &wd[4], 16);
      __builtin_memcpy(&p.Aop[m][2], &hw[0], 16);
    }
  };
  // the matrix products of a hypothesis tile (three MFMAs per vertex, the accumulators start from kap) ...
  auto products = [&](int T, const Prep &p, const float *kap_buf, sgtd_f32x16 (&acc)[3]) {
#pragma unroll
    for (int m = 0; m < 3; m++) {
      sgtd_f32x16 a;
#pragma unroll
      for (int k = 0; k < 4; k++) {        // kap of rows 8 k + 4 hh + (0..3) = registers 4 k + (0..3)
        const float4 kk = *reinterpret_cast<const float4 *>(kap_buf + m * 32 + 8 * k + 4 * (int)hh);
        a[4 * k] = kk.x; a[4 * k + 1] = kk.y; a[4 * k + 2] = kk.z; a[4 * k + 3] = kk.w;
      }
#ifdef SGTD_EXP_VM_NOMFMA
      a[0] += (float)p.Aop[m][0][0] + (float)p.Aop[m][1][7] + (float)Bop[T][0][0];     // (experiment: timing without the matrix pipe)
#else
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(p.Aop[m][0], Bop[T][0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(p.Aop[m][1], Bop[T][1], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(p.Aop[m][2], Bop[T][2], a, 0, 0, 0);
#endif
      acc[m] = a;
    }
  };
  // ... and the look at their results: this lane's vote bits and open bits of the hypothesis tile
  auto look = [&](int T, const Prep &p, const sgtd_f32x16 (&acc)[3], u32 &word, u32 &open) {
    // z = s (d^2 - 9) + e of the worst vertex, e = the bound of its error: z < 0 is a certain vote, z > 2 e certainly none —
    // the signs of z and of z - 2 e, two registers per packed addition
    const float bh = (p.Umax + t1[T]) * (p.Umax + t1[T]) + 16.0f;
    const float eh = __builtin_fmaf(p.se, bh, 0.12f);
    const f32x2 c1 = {eh, eh}, c2 = {-eh * 1.000001f, -eh * 1.000001f};
    u32 sure = 0u, und = 0u;
#pragma unroll
    for (int g = 14; g >= 0; g -= 2) {
      const f32x2 mx = {__builtin_fmaxf(__builtin_fmaxf(acc[0][g], acc[1][g]), acc[2][g]),
                        __builtin_fmaxf(__builtin_fmaxf(acc[0][g + 1], acc[1][g + 1]), acc[2][g + 1])};
      const f32x2 z1 = mx + c1, z2 = mx + c2;
      sure = __builtin_amdgcn_alignbit(sure, __float_as_uint(z1.y), 31);            // (sure << 1) | sign(z)
      sure = __builtin_amdgcn_alignbit(sure, __float_as_uint(z1.x), 31);
      und = __builtin_amdgcn_alignbit(und, __float_as_uint(z2.y), 31);              // (und << 1) | sign(z - 2 e)
      und = __builtin_amdgcn_alignbit(und, __float_as_uint(z2.x), 31);
    }
    VMSTAT(0, 1);
    // rows without a matrix result: beyond the list's end — nothing —, or a wild pair — exact test; a hypothesis the matrix
    // pass could not take: exact test for every pair; a lane without a hypothesis: nothing
    sure &= ~p.off16 & 0xFFFFu;
    und = ((und & ~sure & ~p.off16) | p.wild16) & 0xFFFFu;
    if (hyp_exact[T]) { sure = 0u; und = 0xFFFFu & ~(p.off16 & ~p.wild16); }
    if (!hyp_ok[T]) { sure = 0u; und = 0u; }
    cnt[T] += (u32)__builtin_popcount(sure);
    word |= sure << (16 * T);
    open |= und << (16 * T);
  };
  // queue what the matrix pass left open, densely: a prefix sum of the lanes' counts places every lane's combinations
  // (at most 64 x 16 per hypothesis tile: the queue always has room for one tile's after it has been worked off)
  auto enqueue = [&](u32 tile, u32 open) {
    if (__builtin_amdgcn_ballot_w64(open != 0u)) {
      VMSTAT(1, 1);
#pragma unroll 1
      for (int T = 0; T < 2; T++) {
        u32 m16 = (open >> (16 * T)) & 0xFFFFu;
        const u32 mine = (u32)__builtin_popcount(m16), incl = wave_incl_scan(mine), total = (u32)__builtin_amdgcn_readlane((int)incl, SGTD_WAVE - 1);
        if (!total) continue;
        if (qn + total > SGTD_VM_QCAP) { vm_drain(P, bid, queue, qn, base, qslot0, words, s_votes); qn = 0; VMSTAT(4, 1); }
        u32 pos = qn + incl - mine;
        while (m16) {
          const u32 g = (u32)__builtin_ctz(m16);
          m16 &= m16 - 1u;
          queue[pos++] = ((u64)(tile * 32u + vm_row(g, hh)) << 8) | (u64)(T * 32 + (int)r);
        }
        qn += total;
      }
    }
  };

#if SGTD_VM_PIPE
  // Software pipeline: while tile i's MFMAs run, the vector unit prepares tile i + 1 (its products are independent of them), then
  // looks at tile i's results.  kap alternates between two LDS buffers (tile i's is read while tile i + 1's is written).
  Tile raw;                                    // the vertices of the tile that is prepared next
  vertices_of(pair_of((u32)wave), raw);
  u64 pr_next = pair_of(min((u32)wave + STEP, last_tile));
  Prep cur;
  prepare(raw, (u32)wave, kap_w, cur);
  vertices_of(pr_next, raw);                   // tile wave + STEP (clamped)
  pr_next = pair_of(min((u32)wave + 2 * STEP, last_tile));
  u32 flip = 0;
  for (u32 tile = (u32)wave; tile < n_tiles; tile += STEP) {
    float *kap_cur = kap_w + flip * 96, *kap_nxt = kap_w + (flip ^ 1u) * 96;
    Prep nxt;
    u32 word = 0, open = 0;
#if SGTD_VM_PIPE == 2
    // both hypothesis tiles' eighteen MFMAs with the whole preparation of the next tile between them
    sgtd_f32x16 acc[3], acc1[3];
    __builtin_amdgcn_sched_barrier(0);
    products(0, cur, kap_cur, acc);
    products(1, cur, kap_cur, acc1);
    prepare(raw, tile + STEP, kap_nxt, nxt);
    vertices_of(pr_next, raw);
    pr_next = pair_of(min(tile + 3 * STEP, last_tile));
#pragma unroll
    for (int i = 0; i < 18; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, SGTD_VM_VALU_PER_MFMA / 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    look(0, cur, acc, word, open);
    look(1, cur, acc1, word, open);
    words[(size_t)tile * 64 + lane] = word;
    enqueue(tile, open);
    cur = nxt;
    flip ^= 1u;
    continue;
#else
    sgtd_f32x16 acc[3];
    __builtin_amdgcn_sched_barrier(0);
    products(0, cur, kap_cur, acc);
    prepare(raw, tile + STEP, kap_nxt, nxt);                       // (under tile i's first MFMAs)
    vertices_of(pr_next, raw);                                     // tile i + 2 STEP's vertices, i + 3 STEP's pair word
    pr_next = pair_of(min(tile + 3 * STEP, last_tile));
    // (the order the scheduler is asked for: a matrix instruction, then its share of the preparation's vector instructions — left
    // to itself it issues the nine MFMAs in a row and the wave stands at the matrix pipe while its vector work waits)
#pragma unroll
    for (int i = 0; i < 9; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, SGTD_VM_VALU_PER_MFMA, 0);
    }
    __builtin_amdgcn_sched_barrier(0);       // (the look's vector instructions, which wait for the MFMAs, stay behind this line)
    look(0, cur, acc, word, open);
    if (n_ht > 1) {
      products(1, cur, kap_cur, acc);
      look(1, cur, acc, word, open);
    }
    words[(size_t)tile * 64 + lane] = word;
    enqueue(tile, open);
    cur = nxt;
    flip ^= 1u;
#endif
  }
#else
  Tile cur;
  vertices_of(pair_of((u32)wave), cur);
  u64 pr_next = pair_of(min((u32)wave + STEP, last_tile));
  for (u32 tile = (u32)wave; tile < n_tiles; tile += STEP) {
    Tile nxt;
    vertices_of(pr_next, nxt);
    const u64 pr_after = pair_of(min(tile + 2 * STEP, last_tile));
    Prep p;
    prepare(cur, tile, kap_w, p);
    u32 word = 0, open = 0;
    // both tiles' MFMAs first — the second tile's run on the matrix pipe while the vector unit looks at the first tile's
    // results (96 accumulator registers instead of 48)
    sgtd_f32x16 acc0[3], acc1[3];
    products(0, p, kap_w, acc0);
    products(1, p, kap_w, acc1);
    look(0, p, acc0, word, open);
    look(1, p, acc1, word, open);
    words[(size_t)tile * 64 + lane] = word;
    enqueue(tile, open);
    cur = nxt;
    pr_next = pr_after;
  }
#endif
  if (qn) { vm_drain(P, bid, queue, qn, base, qslot0, words, s_votes); VMSTAT(4, 1); }
#pragma unroll
  for (int T = 0; T < 2; T++)
    if (hyp_ok[T] && cnt[T]) { atomicAdd(&s_votes[T * 32 + (int)r], cnt[T]); VMSTAT1(5, cnt[T]); }
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
    if (tid == 0) { *score = -1.0; if (P.inl_count) P.inl_count[bid] = 0u; }                   // :541
    if (tid < 12) pose[tid] = 0.0;
    for (u32 jj = tid; jj < n; jj += THREADS) P.inlier[base + jj] = 0;
    return;
  }
  // ---- inliers of the best hypothesis (:516-539): the pairs that voted for it (the words were written by other lanes, some
  // bits by atomics: read past the vector cache)
  u32 mine = 0;
  for (u32 jj = tid; jj < n; jj += THREADS) {
    const u32 row = jj & 31u, h2 = (row >> 2) & 1u, g = ((row >> 3) << 2) | (row & 3u);
    const u32 wv = __hip_atomic_load(&words[(size_t)(jj >> 5) * 64 + (best & 31u) + 32u * h2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool in = (wv >> (g + 16u * (best >> 5))) & 1u;
    P.inlier[base + jj] = in ? 1 : 0;
    mine += in ? 1u : 0u;
  }
  mine = wave_sum(mine);
  if (lane == 0 && mine) atomicAdd(&s_count, mine);
  __syncthreads();
  if (tid == 0) { *score = (double)s_count; if (P.inl_count) P.inl_count[bid] = s_count; }          // :539
  if (tid < 12) pose[tid] = P.hyp64[((size_t)bid * SGTD_VERIFY_MAX_HYP + best) * SGTD_HYP_F64 + tid];
}
