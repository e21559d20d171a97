// Does a vector-memory store issued with EXEC = 0 take a place in vmcnt's order on gfx950?
// Each wave starts a load from a cold line, issues N stores with exec = 0 behind it, waits for "at most N operations
// outstanding" and reads the load's destination at once.  If the empty stores are counted, the wait covers the load and
// every lane sees the loaded value; if they are not, the wait falls through and lanes see the register's old value.
//   hipcc --offload-arch=gfx950 -O3 -o build/vmcnt_exec0 tools/vmcnt_exec0.hip && build/vmcnt_exec0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
__global__ void probe(const u32 *src, u32 *dump, u32 *stale, size_t stride) {
  const u32 *p = src + (size_t)blockIdx.x * stride + threadIdx.x;
  u32 v = 0xDEADBEEFu;
  asm volatile("v_mov_b32 %0, 0xDEADBEEF" : "=v"(v));
  asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(p));
  u32 *d = dump + threadIdx.x;
  unsigned long long sv;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0\n\t"
               "global_store_dword %1, %2, off\n\tglobal_store_dword %1, %2, off\n\t"
               "global_store_dword %1, %2, off\n\tglobal_store_dword %1, %2, off\n\t"
               "s_mov_b64 exec, %0\n\ts_waitcnt vmcnt(4)"
               : "=&s"(sv) : "v"(d), "v"(threadIdx.x));
  u32 seen;
  asm volatile("v_mov_b32 %0, %1" : "=v"(seen) : "v"(v));
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
  if (seen != v) atomicAdd(stale, 1u);
}
int main() {
  const int blocks = 4096; const size_t stride = 1 << 16;   // 256 KB apart: every load a cold line
  u32 *src, *dump, *stale;
  hipMalloc(&src, blocks * stride * 4); hipMalloc(&dump, 4096); hipMalloc(&stale, 4);
  hipMemset(src, 0x11, blocks * stride * 4); hipMemset(stale, 0, 4);
  probe<<<blocks, 64>>>(src, dump, stale, stride);
  u32 h = 0; hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost);
  printf("lanes that read their register before the load arrived: %u of %d  => stores with EXEC = 0 are %s by vmcnt\n",
         h, blocks * 64, h ? "NOT counted" : "counted");
  return 0;
}
