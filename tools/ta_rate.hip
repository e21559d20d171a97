// Microbenchmark: vector-memory instruction throughput of one CU (L1/L2-resident data).
// Not part of the product; used to size the sweep kernel's loads (sweep section).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;

template <int W, int STRIDE_LANES>   // W = dwords per lane per load (1,2,4)
__global__ __launch_bounds__(256) void k(const u32 *buf, u32 mask_dwords, int iters, u32 *out, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63;
  const u32 wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  u32 acc = 0;
  u32 base = (wave * 977u) & mask_dwords;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const u32 off = ((base + (u32)(u * 64 * W * STRIDE_LANES)) + (u32)lane * W * STRIDE_LANES) & mask_dwords;
      if (W == 1) acc ^= buf[off];
      else if (W == 2) { uint2 v = *reinterpret_cast<const uint2 *>(buf + (off & ~1u)); acc ^= v.x ^ v.y; }
      else { uint4 v = *reinterpret_cast<const uint4 *>(buf + (off & ~3u)); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    base = (base + 8 * 64 * W * STRIDE_LANES + 4 * (u32)i) & mask_dwords;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345678u) out[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int W, int S>
void run(const char *name, const u32 *buf, u32 mask, u32 *out, unsigned long long *cyc, int wgs_per_cu, int n_cus) {
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<W, S><<<n_cus * wgs_per_cu, 256>>>(buf, mask, 10, out, cyc);
  hipEventRecord(a);
  k<W, S><<<n_cus * wgs_per_cu, 256>>>(buf, mask, iters, out, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double instr_per_cu = (double)wgs_per_cu * 4 * iters * 8;
  printf("%-28s footprint %6u KB  waves/SIMD %d: %.1f cycles/instr/CU (clock %.2f GHz), %.1f B/clk/CU, %.2f TB/s chip\n", name, (mask + 1) / 256,
         wgs_per_cu, (double)c / instr_per_cu, c / (ms * 1e6), 64.0 * 4 * W * instr_per_cu / c, 64.0 * 4 * W * instr_per_cu * n_cus / (ms * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cus = p.multiProcessorCount;
  u32 *buf, *out; unsigned long long *cyc;
  const size_t bytes = 1ull << 30;
  hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 64); hipMalloc(&cyc, 64);
  for (u32 kb : {16u, 2048u, 262144u}) {
    const u32 mask = kb * 256 - 1;
    for (int occ : {4, 8}) {
      run<1, 1>("dword   contiguous", buf, mask, out, cyc, occ, n_cus);
      run<2, 1>("dwordx2 contiguous", buf, mask, out, cyc, occ, n_cus);
      run<4, 1>("dwordx4 contiguous", buf, mask, out, cyc, occ, n_cus);
      run<4, 2>("dwordx4 stride 32 B", buf, mask, out, cyc, occ, n_cus);
    }
  }
  return 0;
}
