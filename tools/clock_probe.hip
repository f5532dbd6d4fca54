// Effective shader clock on MI355X: s_memtime (shader-clock ticks per the guide) vs s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_busy(unsigned long long* out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
    a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[4096] = 1;
  if (threadIdx.x == 0 && blockIdx.x < 1024) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
__global__ void k_stream(const u32x4* src, unsigned* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; unsigned acc = 0;
  for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { u32x4 v = __builtin_nontemporal_load(src + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) out[0] = acc;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 1 << 20);
  char* pool; hipMalloc(&pool, 2ull << 30); hipMemset(pool, 1, 2ull << 30);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto report = [&](const char* tag, float ms, int iters) {
    std::vector<unsigned long long> h(2048); hipMemcpy(h.data(), d, 2048 * 8, hipMemcpyDeviceToHost);
    double mt = 0, rt = 0; for (int i = 0; i < 1024; ++i) { mt += h[2 * i]; rt += h[2 * i + 1]; } mt /= 1024; rt /= 1024;
    printf("%-34s wall %8.2f us | memtime %9.0f ticks, memrealtime %7.0f ticks (=%.2f us @100MHz) | memtime MHz (vs realtime) %7.1f | %.2f memtime ticks per fma-iter(8 fma)\n",
           tag, ms * 1e3, mt, rt, rt / 100.0, mt / (rt / 100.0), mt / iters);
  };
  for (int iters : {200, 2000, 20000}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k_busy, dim3(1024), dim3(512), 0, 0, d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); char tag[64]; snprintf(tag, 64, "busy iters=%d rep%d", iters, rep); report(tag, ms, iters);
    }
  }
  // sustained: 200 launches of streaming + busy interleaved, then measure
  for (int r = 0; r < 200; ++r) { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const u32x4*)pool, (unsigned*)(d + 8192), (size_t)(64 << 20) / 16); hipLaunchKernelGGL(k_busy, dim3(1024), dim3(512), 0, 0, d, 500, 1.0f); }
  hipEventRecord(e0); hipLaunchKernelGGL(k_busy, dim3(1024), dim3(512), 0, 0, d, 2000, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); report("busy iters=2000 after sustained mix", ms, 2000);
  return 0;
}
