// How many atomic adds per second does ONE address take on gfx950 (device scope, executed in L2), with and without a
// returned value, and how does that grow with the number of addresses (4 KB apart)?  The planner's pool cursor, the
// sweep's slab cursor and its ticket heads are such addresses.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_rate tools/atomic_rate.hip && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
template <bool RET>
__global__ void hammer(u32 *ctr, u32 n_addr, int iters, u32 *sink) {
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (threadIdx.x & 63) return;
  u32 *p = ctr + (size_t)(wave % n_addr) * 1024;
  u32 acc = 0;
  for (int i = 0; i < iters; i++) {
    if (RET) acc += atomicAdd(p, 1u + (acc & 1u));     // (the next add depends on the value returned)
    else atomicAdd(p, 1u);
  }
  if (RET && acc == 0xFFFFFFFFu) *sink = acc;
}
int main() {
  u32 *ctr, *sink;
  hipMalloc(&ctr, 4096 * 1024); hipMalloc(&sink, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 2048, threads = 256, iters = 64;       // 8192 waves, one lane each
  for (int ret = 1; ret >= 0; ret--)
    for (u32 n_addr : {1u, 8u, 64u, 512u}) {
      hipMemset(ctr, 0, 4096 * 1024);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (ret) hammer<true><<<blocks, threads>>>(ctr, n_addr, iters, sink); else hammer<false><<<blocks, threads>>>(ctr, n_addr, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      const double n = (double)blocks * threads / 64 * iters;
      printf("%s, %3u address(es): %.0f atomics in %.3f ms = %.1f M/s per address, %.1f ns each\n", ret ? "returning" : "no return", n_addr, n, ms,
             n / ms / 1e3 / n_addr, ms * 1e6 / (n / n_addr));
    }
  return 0;
}
