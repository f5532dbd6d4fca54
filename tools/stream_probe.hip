// Streaming-read floor on MI355X for GEMV-sized buffers (cold HBM, rotating through 4 GiB).
// Reports per-launch dispatch duration (hipExtLaunchKernelGGL events) and back-to-back wall time.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(unsigned* out) { if (out == nullptr) out[0] = 1; }

// each wave reads BPW consecutive 1 KiB blocks (all loads issued before the first use)
template <int BPW, bool NT>
__global__ __launch_bounds__(1024) void k_read(const u32x4* __restrict__ src, unsigned* out, long nblocks_total) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const u32x4* p = src + wave * BPW * 64 + lane;
  u32x4 v[BPW];
#pragma unroll
  for (int i = 0; i < BPW; ++i) v[i] = NT ? __builtin_nontemporal_load(p + i * 64) : p[i * 64];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < BPW; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (acc == 0x12345678u) out[0] = acc;
}

template <int BPW, bool NT>
int run(const char* tag, const char* base, size_t pool, size_t bytes, int wpb, unsigned* dout, hipStream_t s) {
  const long blocks = bytes / 1024;
  const long waves = blocks / BPW;
  const int grid = (int)(waves / wpb);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ev;
  size_t off = 0;
  for (int it = 0; it < 60; ++it) {
    off = (off + bytes + (64 << 20)) % (pool - bytes); off &= ~(size_t)4095;
    hipExtLaunchKernelGGL((k_read<BPW, NT>), dim3(grid), dim3(wpb * 64), 0, s, e0, e1, 0, (const u32x4*)(base + off), dout, blocks);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 10) ev.push_back(ms * 1e3f);
  }
  std::sort(ev.begin(), ev.end());
  // hot: same buffer repeatedly (Infinity Cache / TLB warm)
  std::vector<float> hot;
  for (int it = 0; it < 30; ++it) {
    hipExtLaunchKernelGGL((k_read<BPW, NT>), dim3(grid), dim3(wpb * 64), 0, s, e0, e1, 0, (const u32x4*)base, dout, blocks);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 10) hot.push_back(ms * 1e3f);
  }
  std::sort(hot.begin(), hot.end());
  // back-to-back wall (cold, rotating)
  const int NL = 200;
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int it = 0; it < NL; ++it) {
    off = (off + bytes + (64 << 20)) % (pool - bytes); off &= ~(size_t)4095;
    hipLaunchKernelGGL((k_read<BPW, NT>), dim3(grid), dim3(wpb * 64), 0, s, (const u32x4*)(base + off), dout, blocks);
  }
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float wall; CK(hipEventElapsedTime(&wall, e0, e1));
  const float med = ev[ev.size() / 2], hmed = hot[hot.size() / 2], b2b = wall * 1e3f / NL;
  printf("%-6s %6.1f MB bpw=%2d wpb=%2d grid=%5d nt=%d | cold ev med %6.2f us (%5.0f GB/s) min %6.2f | hot %6.2f us (%5.0f GB/s) | b2b %6.2f us (%5.0f GB/s)\n",
         tag, bytes / 1e6, BPW, wpb, grid, (int)NT, med, bytes / med / 1e3, ev[0], hmed, bytes / hmed / 1e3, b2b, bytes / b2b / 1e3);
  return 0;
}

int main() {
  const size_t pool = (size_t)4 << 30;
  char* base; CK(hipMalloc(&base, pool)); CK(hipMemset(base, 1, pool));
  unsigned* dout; CK(hipMalloc(&dout, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  // empty kernel
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int g : {1, 256, 1024}) {
      std::vector<float> ev;
      for (int it = 0; it < 40; ++it) {
        hipExtLaunchKernelGGL(k_empty, dim3(g), dim3(512), 0, s, e0, e1, 0, dout);
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 10) ev.push_back(ms * 1e3f);
      }
      std::sort(ev.begin(), ev.end());
      CK(hipEventRecord(e0, s));
      for (int it = 0; it < 200; ++it) hipLaunchKernelGGL(k_empty, dim3(g), dim3(512), 0, s, dout);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float wall; CK(hipEventElapsedTime(&wall, e0, e1));
      printf("empty kernel grid=%4d x512: event med %.2f us min %.2f; back-to-back %.2f us/launch\n", g, ev[ev.size() / 2], ev[0], wall * 1e3 / 200);
    }
  }
  const size_t o = 4096ull * 4096 / 2, qkv = 6144ull * 4096 / 2, gate = 14336ull * 4096 / 2;
  for (size_t bytes : {o, qkv, gate}) {
    const char* tag = bytes == o ? "o" : (bytes == qkv ? "qkv" : "gate");
    run<4, true>(tag, base, pool, bytes, 8, dout, s);
    run<4, false>(tag, base, pool, bytes, 8, dout, s);
    run<4, true>(tag, base, pool, bytes, 4, dout, s);
    run<4, true>(tag, base, pool, bytes, 16, dout, s);
    run<8, true>(tag, base, pool, bytes, 4, dout, s);
    run<8, true>(tag, base, pool, bytes, 8, dout, s);
    run<16, true>(tag, base, pool, bytes, 4, dout, s);
    run<16, true>(tag, base, pool, bytes, 8, dout, s);
    run<2, true>(tag, base, pool, bytes, 16, dout, s);
    run<1, true>(tag, base, pool, bytes, 16, dout, s);
  }
  return 0;
}
