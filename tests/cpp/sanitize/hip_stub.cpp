// hip_stub.cpp — a malloc-backed stand-in for the HIP runtime, for ONE purpose: running the engine's HOST code
// (sgtd_amd/csrc/sgtd_accel.hip + multi_impl.hip.h: buffer growth, room bookkeeping, page-locked staging, re-runs, views,
// table files) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (no GPU sanitizer exists on
// the pool).  The engine is compiled for the host only (hipcc --cuda-host-only -fsanitize=address,undefined) and linked
// against this file instead of libamdhip64:
//   device memory   = zero-initialised host memory: every copy the engine makes into or out of a "device" buffer is a
//                     memcpy the sanitizer checks against the allocation's real size
//   kernels         = nothing runs; a launch calls the test's hook with the kernel's name and argument array, so a
//                     scenario can leave behind what a kernel would have (overflow flags, counts) and drive the host
//                     through its growth and re-run paths
//   streams, events = everything is synchronous and complete
// Test infrastructure only: nothing in the product links it.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
// (function-local statics: the engine's module constructor registers its kernels before this file's globals would exist)
std::mutex &mu() { static std::mutex m; return m; }
std::map<const void *, std::string> &kernels() { static std::map<const void *, std::string> k; return k; }      // host stub address -> device name
std::map<void *, size_t> &blocks() { static std::map<void *, size_t> b; return b; }                             // "device" allocations
std::map<void *, size_t> &host_blocks() { static std::map<void *, size_t> b; return b; }                        // page-locked host allocations
#define g_host_blocks host_blocks()
#define g_mu mu()
#define g_kernels kernels()
#define g_blocks blocks()
size_t g_allocated = 0, g_peak = 0;
unsigned long long g_launches = 0;
typedef void (*launch_hook_t)(const char *name, void **args, void *user);
launch_hook_t g_hook = nullptr;
void *g_hook_user = nullptr;
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local CallCfg t_cfg;
struct Event { double t; };
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr size_t kTotalMem = 6ull << 30, kReserve = 2ull << 30;     // small on purpose: the engine sizes work buffers by what is free
}  // namespace

extern "C" {
// ---- the test's side
void sgtd_stub_set_launch_hook(launch_hook_t h, void *user) { g_hook = h; g_hook_user = user; }
unsigned long long sgtd_stub_launches() { return g_launches; }
size_t sgtd_stub_device_bytes() { return g_allocated; }
size_t sgtd_stub_device_peak() { return g_peak; }
size_t sgtd_stub_device_blocks() { return g_blocks.size(); }
size_t sgtd_stub_block_size(const void *p) {           // bytes of the allocation that starts at p (0: not one)
  std::lock_guard<std::mutex> l(g_mu);
  auto it = g_blocks.find(const_cast<void *>(p));
  if (it != g_blocks.end()) return it->second;
  it = g_host_blocks.find(const_cast<void *>(p));
  return it == g_host_blocks.end() ? 0 : it->second;
}

// ---- what hipcc's host code calls
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *host_fn, char *, const char *device_name, unsigned int, void *, void *, void *, void *, int *) {
  std::lock_guard<std::mutex> l(g_mu);
  g_kernels[host_fn] = device_name ? device_name : "?";
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  t_cfg = CallCfg{grid, block, shmem, stream};
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
  *grid = t_cfg.grid; *block = t_cfg.block; *shmem = t_cfg.shmem; *stream = t_cfg.stream;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t) {
  std::string name;
  {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_kernels.find(fn);
    name = it == g_kernels.end() ? "?" : it->second;
    g_launches++;
  }
  // what a real launch would refuse
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 || shmem > 160 * 1024) {
    fprintf(stderr, "hip_stub: invalid launch of %s: grid %u %u %u block %u %u %u shmem %zu\n", name.c_str(), grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
    return hipErrorInvalidConfiguration;
  }
  if (g_hook) g_hook(name.c_str(), args, g_hook_user);
  return hipSuccess;
}

// ---- devices
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int d) {
  if (d != 0) return hipErrorInvalidDevice;
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "host stand-in");
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256; p->totalGlobalMem = kTotalMem; p->warpSize = 64; p->maxThreadsPerBlock = 1024;
  p->sharedMemPerBlock = 160 * 1024; p->maxSharedMemoryPerMultiProcessor = 160 * 1024; p->clockRate = 2400000;
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  *total_b = kTotalMem;
  *free_b = g_allocated + kReserve < kTotalMem ? kTotalMem - kReserve - g_allocated : 0;
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
hipError_t hipFuncGetAttributes(hipFuncAttributes *a, const void *) { memset(a, 0, sizeof(*a)); a->sharedSizeBytes = 12928; a->maxThreadsPerBlock = 1024; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int value) { return value <= 160 * 1024 ? hipSuccess : hipErrorInvalidValue; }

// ---- memory
hipError_t hipMalloc(void **p, size_t n) {
  if (g_allocated + n > kTotalMem) { *p = nullptr; return hipErrorOutOfMemory; }
  void *q = calloc(n ? n : 1, 1);
  if (!q) { *p = nullptr; return hipErrorOutOfMemory; }
  std::lock_guard<std::mutex> l(g_mu);
  g_blocks[q] = n; g_allocated += n; g_peak = g_allocated > g_peak ? g_allocated : g_peak;
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_blocks.find(p);
    if (it == g_blocks.end()) { fprintf(stderr, "hip_stub: hipFree of a pointer that is not an allocation: %p\n", p); abort(); }
    g_allocated -= it->second;
    g_blocks.erase(it);
  }
  free(p);
  return hipSuccess;
}
// (page-locked host memory is device-visible: a kernel may be handed it, so its blocks are known by size too)
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) {
  *p = calloc(n ? n : 1, 1);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> l(g_mu);
  g_host_blocks[*p] = n;
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_host_blocks.find(p);
    if (it == g_host_blocks.end()) { fprintf(stderr, "hip_stub: hipHostFree of a pointer that is not an allocation: %p\n", p); abort(); }
    g_host_blocks.erase(it);
  }
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }

// (what the engine asks before it lets a kernel write a caller's array: page-locked blocks are device-visible at their own address)
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
  std::lock_guard<std::mutex> l(g_mu);
  for (auto *m : {&g_host_blocks, &g_blocks}) {
    auto it = m->upper_bound(const_cast<void *>(p));
    if (it == m->begin()) continue;
    --it;
    if (static_cast<const char *>(p) < static_cast<const char *>(it->first) + it->second) {
      memset(a, 0, sizeof(*a));
      a->type = m == &g_host_blocks ? hipMemoryTypeHost : hipMemoryTypeDevice;
      a->devicePointer = const_cast<void *>(p); a->hostPointer = m == &g_host_blocks ? const_cast<void *>(p) : nullptr;
      return hipSuccess;
    }
  }
  return hipErrorInvalidValue;
}
hipError_t hipMemGetAddressRange(hipDeviceptr_t *base, size_t *size, hipDeviceptr_t p) {
  std::lock_guard<std::mutex> l(g_mu);
  for (auto *m : {&g_host_blocks, &g_blocks}) {
    auto it = m->upper_bound(p);
    if (it == m->begin()) continue;
    --it;
    if (static_cast<const char *>(p) < static_cast<const char *>(it->first) + it->second) { *base = it->first; *size = it->second; return hipSuccess; }
  }
  return hipErrorInvalidValue;
}

// ---- streams and events
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(new Event{0}); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<Event *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { reinterpret_cast<Event *>(e)->t = now_ms(); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(reinterpret_cast<Event *>(b)->t - reinterpret_cast<Event *>(a)->t); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
}
