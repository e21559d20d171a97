// pinned_probe.hip — how fast is page-locked host memory on this box, from both sides?  For every way of getting it
// (hipHostMalloc Default / NonCoherent / Coherent / WriteCombined, hipHostRegister of malloc'd memory): a kernel writes 576 KB into it
// (16 bytes per lane, coalesced), the host copies the 576 KB out with memcpy, and a hipMemcpyAsync device -> it is timed beside.
//   hipcc --offload-arch=gfx950 -O2 -o variants/pinned_probe tools/pinned_probe.hip && ./variants/pinned_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill_wide(uint4 *out, size_t n16, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}
__global__ void fill_narrow(double *out, size_t n8, unsigned v) {      // 8-byte pieces, stride 24 bytes between lanes (one field of three doubles per lane)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8 / 3; i += (size_t)gridDim.x * blockDim.x)
    for (int k = 0; k < 3; k++) out[i * 3 + k] = (double)(v + k);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 576 * 1024;
  void *dev = nullptr;
  CK(hipMalloc(&dev, bytes));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<char> dst(bytes), src(bytes, 1);
  struct Kind { const char *name; unsigned flags; bool reg; } kinds[] = {
      {"hipHostMalloc Default", hipHostMallocDefault, false}, {"hipHostMalloc NonCoherent", hipHostMallocNonCoherent, false},
      {"hipHostMalloc Coherent", hipHostMallocCoherent, false}, {"hipHostMalloc WriteCombined", hipHostMallocWriteCombined, false},
      {"hipHostMalloc Portable|Mapped", hipHostMallocPortable | hipHostMallocMapped, false}, {"hipHostRegister(malloc)", 0, true}};
  for (const Kind &k : kinds) {
    void *h = nullptr, *raw = nullptr;
    if (k.reg) { raw = aligned_alloc(4096, bytes); memset(raw, 0, bytes); if (hipHostRegister(raw, bytes, hipHostRegisterDefault) != hipSuccess) { printf("%s: refused\n", k.name); (void)hipGetLastError(); free(raw); continue; } h = raw; }
    else if (hipHostMalloc(&h, bytes, k.flags) != hipSuccess) { printf("%s: refused\n", k.name); (void)hipGetLastError(); continue; }
    void *hd = h;
    if (k.reg) CK(hipHostGetDevicePointer(&hd, h, 0));
    double t_wide = 1e30, t_narrow = 1e30, t_copy = 1e30, t_read = 1e30, t_write = 1e30, t_h2d = 1e30;
    for (int it = 0; it < 6; it++) {
      float ms;
      CK(hipEventRecord(a, s)); fill_wide<<<64, 256, 0, s>>>((uint4 *)hd, bytes / 16, it); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b)); t_wide = ms * 1e3 < t_wide ? ms * 1e3 : t_wide;
      double t0 = now_us(); memcpy(dst.data(), h, bytes); double t1 = now_us(); t_read = t1 - t0 < t_read ? t1 - t0 : t_read;
      if (((unsigned *)dst.data())[0] != (unsigned)it) { printf("%s: the host does not see the kernel's writes\n", k.name); }
      CK(hipEventRecord(a, s)); fill_narrow<<<64, 256, 0, s>>>((double *)hd, bytes / 8, it); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b)); t_narrow = ms * 1e3 < t_narrow ? ms * 1e3 : t_narrow;
      t0 = now_us(); CK(hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t1 = now_us(); t_copy = t1 - t0 < t_copy ? t1 - t0 : t_copy;
      t0 = now_us(); memcpy(h, src.data(), bytes); t1 = now_us(); t_write = t1 - t0 < t_write ? t1 - t0 : t_write;
      t0 = now_us(); CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t1 = now_us(); t_h2d = t1 - t0 < t_h2d ? t1 - t0 : t_h2d;
    }
    printf("%-30s kernel writes 576 KB: wide %.1f us, 8-byte pieces %.1f us | host memcpy out of it %.1f us, into it %.1f us | hipMemcpyAsync d2h + wait %.1f us, h2d + wait %.1f us\n", k.name, t_wide, t_narrow, t_read, t_write, t_copy, t_h2d);
    if (k.reg) { CK(hipHostUnregister(raw)); free(raw); } else CK(hipHostFree(h));
  }
  // what the runtime says about page-locked memory (sgtd_search_frame asks before it lets a kernel write a caller's array)
  { void *h = nullptr; CK(hipHostMalloc(&h, bytes, hipHostMallocPortable));
    for (size_t off : {(size_t)0, (size_t)4096, bytes - 8}) {
      hipPointerAttribute_t a; memset(&a, 0, sizeof a);
      hipError_t r1 = hipPointerGetAttributes(&a, (char *)h + off);
      hipDeviceptr_t base = nullptr; size_t size = 0;
      hipError_t r2 = hipMemGetAddressRange(&base, &size, (char *)h + off);
      printf("page-locked + %zu: hipPointerGetAttributes %d type %d devicePointer %s hostPointer %s isManaged %d | hipMemGetAddressRange %d base %s size %zu\n", off, (int)r1, (int)a.type,
             a.devicePointer == (char *)h + off ? "same" : "other", a.hostPointer == (char *)h + off ? "same" : "other", (int)a.isManaged, (int)r2, base == h ? "block" : "other", size);
      (void)hipGetLastError();
    }
    std::vector<char> plain(4096);
    hipPointerAttribute_t a; hipError_t r1 = hipPointerGetAttributes(&a, plain.data()); (void)hipGetLastError();
    printf("ordinary memory: hipPointerGetAttributes %d type %d\n", (int)r1, (int)a.type);
    CK(hipHostFree(h)); }
  // eight copies of 72 KB queued back to back, one wait (what copy_out did)
  { void *h = nullptr; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    double best = 1e30;
    for (int it = 0; it < 6; it++) { double t0 = now_us(); for (int c = 0; c < 8; c++) CK(hipMemcpyAsync((char *)h + c * 73728, (char *)dev + c * 73728, 73728, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t1 = now_us(); best = t1 - t0 < best ? t1 - t0 : best; }
    printf("eight hipMemcpyAsync d2h of 72 KB + one wait: %.1f us\n", best); CK(hipHostFree(h)); }
  { double t0 = now_us(); memcpy(dst.data(), src.data(), bytes); double t1 = now_us(); printf("host memcpy of 576 KB between ordinary buffers: %.1f us\n", t1 - t0); }
  return 0;
}
