// PCIe / host-memory probe for the host-resident entry (NMOD_MEM_HOST): what the box gives for
//   pinned H2D / D2H (the roofline of that path), pageable hipMemcpy, hipHostRegister, host memcpy by T threads
//   into a pinned bounce buffer, duplex traffic, and a kernel reading pinned host memory directly.
// Build: hipcc --offload-arch=gfx950 -O2 -pthread tools/pcie_probe.hip -o tools/pcie_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static void par_memcpy(char* dst, const char* src, size_t bytes, int T) {
  if (T <= 1) { memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t per = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; ++t) {
    const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
    if (lo >= bytes) break;
    th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& x : th) x.join();
}

int main(int argc, char** argv) {
  const size_t GB = (size_t)1 << 30, MB = (size_t)1 << 20;
  const size_t big = argc > 1 ? (size_t)atol(argv[1]) * MB : GB;
  CK(hipSetDevice(0));
  printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
  char *dev, *dev2;
  CK(hipMalloc(&dev, big)); CK(hipMalloc(&dev2, big));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // pinned allocation cost
  char* pin;
  double t0 = now();
  CK(hipHostMalloc(&pin, big, hipHostMallocDefault));
  double t1 = now();
  printf("hipHostMalloc %zu MB: %.1f ms\n", big / MB, (t1 - t0) * 1e3);
  t0 = now(); memset(pin, 1, big); t1 = now();
  printf("first touch (memset) of it: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, big / (t1 - t0) / 1e9);
  char* pin2;
  CK(hipHostMalloc(&pin2, big, hipHostMallocDefault));
  memset(pin2, 2, big);

  // pinned H2D / D2H by size
  for (size_t sz : {MB, 4 * MB, 16 * MB, 64 * MB, 256 * MB, big}) {
    if (sz > big) continue;
    float best_h2d = 1e9, best_d2h = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      float ms;
      CK(hipEventRecord(e0, s1)); CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s1)); CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_h2d) best_h2d = ms;
      CK(hipEventRecord(e0, s1)); CK(hipMemcpyAsync(pin2, dev, sz, hipMemcpyDeviceToHost, s1)); CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_d2h) best_d2h = ms;
    }
    printf("pinned %5zu MB: H2D %.2f GB/s  D2H %.2f GB/s\n", sz / MB, sz / (best_h2d * 1e-3) / 1e9, sz / (best_d2h * 1e-3) / 1e9);
  }
  // back-to-back chunks on one stream (what a pipeline issues): 64 x 16 MB, 16 x 64 MB
  for (size_t sz : {4 * MB, 16 * MB, 64 * MB}) {
    const int n = (int)(big / sz);
    t0 = now();
    for (int i = 0; i < n; ++i) CK(hipMemcpyAsync(dev + (size_t)i * sz, pin + (size_t)i * sz, sz, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    t1 = now();
    printf("pinned H2D %d x %zu MB back to back: %.2f GB/s\n", n, sz / MB, big / (t1 - t0) / 1e9);
  }
  // duplex
  {
    t0 = now();
    CK(hipMemcpyAsync(dev, pin, big, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(pin2, dev2, big, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    t1 = now();
    printf("duplex H2D + D2H of %zu MB each: %.1f ms => %.2f GB/s each way\n", big / MB, (t1 - t0) * 1e3, big / (t1 - t0) / 1e9);
  }
  // pageable
  char* pg = (char*)malloc(big);
  memset(pg, 3, big);
  for (int rep = 0; rep < 2; ++rep) {
    t0 = now(); CK(hipMemcpy(dev, pg, big, hipMemcpyHostToDevice)); t1 = now();
    printf("pageable hipMemcpy H2D %zu MB: %.2f GB/s\n", big / MB, big / (t1 - t0) / 1e9);
    t0 = now(); CK(hipMemcpy(pg, dev, big, hipMemcpyDeviceToHost)); t1 = now();
    printf("pageable hipMemcpy D2H %zu MB: %.2f GB/s\n", big / MB, big / (t1 - t0) / 1e9);
  }
  // pageable async on a stream while another stream copies pinned: does it block / overlap?
  // hipHostRegister
  for (size_t sz : {64 * MB, 256 * MB, big}) {
    if (sz > big) continue;
    t0 = now(); CK(hipHostRegister(pg, sz, hipHostRegisterDefault)); t1 = now();
    const double treg = t1 - t0;
    float ms;
    CK(hipEventRecord(e0, s1)); CK(hipMemcpyAsync(dev, pg, sz, hipMemcpyHostToDevice, s1)); CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    t0 = now(); CK(hipHostUnregister(pg)); t1 = now();
    printf("hipHostRegister %5zu MB: %.1f ms (%.1f GB/s), H2D from it %.2f GB/s, unregister %.1f ms\n", sz / MB, treg * 1e3, sz / treg / 1e9,
           sz / (ms * 1e-3) / 1e9, (t1 - t0) * 1e3);
  }
  // host memcpy pageable -> pinned by T threads
  for (int T : {1, 2, 4, 8, 12, 16, 24, 32}) {
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) { t0 = now(); par_memcpy(pin, pg, big, T); t1 = now(); if (t1 - t0 < best) best = t1 - t0; }
    printf("host memcpy pageable -> pinned, %2d threads: %.2f GB/s\n", T, big / best / 1e9);
  }
  // the same while a pinned H2D runs beside it (memory-bandwidth contention)
  for (int T : {4, 8, 16}) {
    t0 = now();
    CK(hipMemcpyAsync(dev, pin2, big, hipMemcpyHostToDevice, s1));
    par_memcpy(pin, pg, big, T);
    t1 = now();
    CK(hipStreamSynchronize(s1));
    double t2 = now();
    printf("memcpy with %2d threads beside a pinned H2D: memcpy %.2f GB/s, both done after %.1f ms (%.2f GB/s H2D)\n", T, big / (t1 - t0) / 1e9, (t2 - t0) * 1e3, big / (t2 - t0) / 1e9);
  }
  // kernel reading pinned host memory directly (zero copy)
  {
    float ms;
    for (int blocks : {256, 1024, 4096}) {
      CK(hipEventRecord(e0, s1));
      hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s1, (const float4*)pin, (float4*)dev, big / 16);
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("kernel copy pinned host -> device, %d blocks: %.2f GB/s\n", blocks, big / (ms * 1e-3) / 1e9);
    }
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, s1, (const float4*)dev2, (float4*)pin2, big / 16);
    CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("kernel copy device -> pinned host: %.2f GB/s\n", big / (ms * 1e-3) / 1e9);
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, s1, (const float4*)dev2, (float4*)dev, big / 16);
    CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("kernel copy device -> device: %.2f GB/s (x2 traffic)\n", big / (ms * 1e-3) / 1e9);
  }
  // stream-ordered pool allocation cost
  {
    void* p;
    t0 = now(); CK(hipMalloc(&p, 256 * MB)); t1 = now();
    double t2 = now(); CK(hipFree(p)); double t3 = now();
    printf("hipMalloc 256 MB: %.2f ms, hipFree %.2f ms\n", (t1 - t0) * 1e3, (t3 - t2) * 1e3);
  }
  return 0;
}
