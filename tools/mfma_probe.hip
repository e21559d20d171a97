// tools/mfma_probe.hip — pins what the verification's MFMA pre-test relies on, on the device itself:
//   * the lane maps of __builtin_amdgcn_mfma_f32_32x32x16_f16 (A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31],
//     D: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5))
//   * that f16 subnormal inputs are NOT flushed (the low parts of the split operands may be subnormal)
//   * that a product of two f16 values enters the f32 sum exactly (a sum of 16 products against an f64 reference)
// build: hipcc -O2 --offload-arch=gfx950 -o build/mfma_probe tools/mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16 *A /*[32][16]*/, const _Float16 *B /*[16][32]*/, float *D /*[32][32]*/) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  h8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int g = 0; g < 16; g++) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
// ---- how the 16 products and C are summed: one lane-0 experiment per case, A row 0 x B column 0 + C[0][0]
__global__ void k_round(const _Float16 *A16 /*[n][16]*/, const _Float16 *B16 /*[n][16]*/, const float *C0 /*[n]*/, float *D0 /*[n]*/, int n) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  for (int t = 0; t < n; t++) {
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = r == 0 ? A16[t * 16 + 8 * h + j] : (_Float16)0.0f; b[j] = r == 0 ? B16[t * 16 + 8 * h + j] : (_Float16)0.0f; }
    f16v c = {0};
    if (l == 0) c[0] = C0[t];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (l == 0) D0[t] = c[0];
  }
}
static void rounding_cases() {
  // case: C, products p_k = a_k b_k.  ulp(1.0) = 2^-23; half ulp = 2^-24
  struct Case { const char *what; float C; int np; float a[16], b[16]; };
  Case cs[8] = {};
  int n = 0;
  auto two = [](int e) { return ldexpf(1.0f, e); };
  { Case &c = cs[n++]; c.what = "C = 1, one product 2^-24 (1 + 2^-6): above half an ulp of C          -> RNE 1 + 2^-23, truncation 1"; c.C = 1.0f; c.np = 1; c.a[0] = two(-12); c.b[0] = two(-12) * (1 + two(-6)); }
  { Case &c = cs[n++]; c.what = "C = 1, one product 2^-24 exactly (a tie)                              -> RNE (to even) 1, round-half-up 1 + 2^-23"; c.C = 1.0f; c.np = 1; c.a[0] = two(-12); c.b[0] = two(-12); }
  { Case &c = cs[n++]; c.what = "C = 1, three products of 2^-26 + one of 2^-25 (together 2^-24 (1 + 1/4)) -> summed exactly first: 1 + 2^-23; added one by one (RNE): 1";
    c.C = 1.0f; c.np = 4; for (int k = 0; k < 3; k++) { c.a[k] = two(-13); c.b[k] = two(-13); } c.a[3] = two(-13); c.b[3] = two(-12); }
  { Case &c = cs[n++]; c.what = "C = 1, sixteen products of 2^-27 (together 2^-23)                      -> summed exactly first: 1 + 2^-23; one by one: 1";
    c.C = 1.0f; c.np = 16; for (int k = 0; k < 16; k++) { c.a[k] = two(-14); c.b[k] = two(-13); } }
  { Case &c = cs[n++]; c.what = "C = 0, products 1 and 2^-24 (1 + 2^-6) and -1                          -> exact sum 2^-24 (1 + 2^-6) = 6.05e-08; left to right with rounding: 0 or 1.19e-07";
    c.C = 0.0f; c.np = 3; c.a[0] = 1.0f; c.b[0] = 1.0f; c.a[1] = two(-12); c.b[1] = two(-12) * (1 + two(-6)); c.a[2] = -1.0f; c.b[2] = 1.0f; }
  { Case &c = cs[n++]; c.what = "C = -1, products 1 (k = 0) and 2^-24 (1 + 2^-6) (k = 8: the other lane half) -> exact 6.05e-08";
    c.C = -1.0f; c.np = 9; c.a[0] = 1.0f; c.b[0] = 1.0f; c.a[8] = two(-12); c.b[8] = two(-12) * (1 + two(-6)); }
  _Float16 hA[8 * 16], hB[8 * 16]; float hC[8], hD[8];
  for (int t = 0; t < n; t++) { hC[t] = cs[t].C; for (int k = 0; k < 16; k++) { hA[t * 16 + k] = (_Float16)cs[t].a[k]; hB[t * 16 + k] = (_Float16)cs[t].b[k]; } }
  _Float16 *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
  k_round<<<1, 64>>>(dA, dB, dC, dD, n);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  for (int t = 0; t < n; t++) printf("rounding: %s\n          got %.9g (C + %.3g ulp of 1)\n", cs[t].what, hD[t], (hD[t] - cs[t].C) / ldexp(1.0, -23));
}

int main() {
  rounding_cases();
  _Float16 hA[32 * 16], hB[16 * 32];
  srand(7);
  for (int i = 0; i < 32 * 16; i++) hA[i] = (_Float16)((rand() % 2001 - 1000) / 64.0f);
  for (int i = 0; i < 16 * 32; i++) hB[i] = (_Float16)((rand() % 2001 - 1000) / 512.0f);
  // row 0 of A x column 0 of B: subnormal f16 (3 * 2^-24) times 32768 = 0.005859375 exactly if not flushed
  for (int kk = 0; kk < 16; kk++) { hA[0 * 16 + kk] = (_Float16)0.0f; hB[kk * 32 + 0] = (_Float16)0.0f; }
  unsigned short sub = 3; memcpy(&hA[0], &sub, 2); hB[0] = (_Float16)32768.0f;
  // row 1 x column 1: subnormal on the B side
  for (int kk = 0; kk < 16; kk++) { hA[1 * 16 + kk] = (_Float16)0.0f; hB[kk * 32 + 1] = (_Float16)0.0f; }
  hA[1 * 16 + 5] = (_Float16)16384.0f; unsigned short sub2 = 5; memcpy(&hB[5 * 32 + 1], &sub2, 2);
  _Float16 *dA, *dB; float *dD; float hD[32 * 32];
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  if (hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
  double worst = 0; int bad = 0;
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
    double ref = 0, mag = 0;
    for (int kk = 0; kk < 16; kk++) { const double p = (double)hA[i * 16 + kk] * (double)hB[kk * 32 + j]; ref += p; mag += fabs(p); }
    const double err = fabs(hD[i * 32 + j] - ref);
    if (mag > 0 && err / mag > worst) worst = err / mag;
    if (err > 1e-6 * (mag + 1e-30)) bad++;
  }
  printf("layout: %d of 1024 elements off (worst |err| / sum|a b| = %.3g, 2^-24 = %.3g)\n", bad, worst, ldexp(1.0, -24));
  printf("subnormal A input: D[0][0] = %.10g (expected %.10g if kept, 0 if flushed)\n", hD[0], 3 * ldexp(1.0, -24) * 32768.0);
  printf("subnormal B input: D[1][1] = %.10g (expected %.10g if kept, 0 if flushed)\n", hD[1 * 32 + 1], 5 * ldexp(1.0, -24) * 16384.0);
  return bad ? 1 : 0;
}
