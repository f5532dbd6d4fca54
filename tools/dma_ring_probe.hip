// What bounds a workgroup-shared LDS-DMA ring on MI355X?  The batched kernels (int4_mm_rb_kernel, rb8_kernel) stage an activation
// tile [rows][128 or 256 B] per k-step into a 3-stage LDS ring (2 steps ahead) from L2 and wait for it with a counted vmcnt + a
// barrier.  This probe runs ONLY that: W workgroups of 4 waves, each streaming `steps` tiles of `tile` bytes from a buffer that
// fits L2 / the Infinity Cache (the activation matrix, re-read by every workgroup), `ahead` steps in flight, with a configurable
// number of spin cycles of "compute" per step.  Output: cycles per step per workgroup and the aggregate DMA rate, for 64 .. 512
// workgroups.  If cycles per step ~ latency / ahead and independent of the compute, the ring depth is the bound.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dma_ring_probe tools/dma_ring_probe.hip && ./dma_ring_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void dma_b128(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TILE_KB per step per workgroup, DPW = DMAs per wave per step (TILE_KB = 4 waves x DPW x 1 KiB), AHEAD steps in flight
template <int DPW, int AHEAD, int SWZ = 0>
__global__ __launch_bounds__(256) void ring_kernel(const char* __restrict__ src, size_t row_bytes, int steps, int spin, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGES = AHEAD + 1;
  constexpr int TILE = 4 * DPW * 1024;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)smem;
  // rows of 128 B: DMA i of wave w fetches rows 8 (DPW w + i) .. + 7 of the step's tile, row stride row_bytes (like the kernels)
  unsigned off[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    // SWZ (round 5): the source-side swizzles the kernels use so that their fragment reads are bank-conflict free -- does permuting the
    // 16-byte pieces of a line among its 8 lanes cost the address coalescer?  1: rb8 / p8 (chunk ^ (row >> 1)); 2: gemm8_dma
    // (chunk ^ row ^ (row >> 3)); 3: halves of the line swapped only (64-byte runs stay); 4: 32-byte runs stay
    const int row = 8 * (DPW * wave + i) + (lane >> 3);
    int c = lane & 7;
    if (SWZ == 1) c ^= (row >> 1) & 7;
    if (SWZ == 2) c ^= (row & 7) ^ ((row >> 3) & 7);
    if (SWZ == 3) c ^= ((row >> 1) & 1) << 2;
    if (SWZ == 4) c ^= ((row >> 1) & 3) << 1;
    off[i] = (unsigned)(row * row_bytes + c * 16);
  }
  auto issue = [&](int stage, int k) {
    const int kk = k < steps ? k : steps - 1;
#pragma unroll
    for (int i = 0; i < DPW; ++i) dma_b128(src + (size_t)kk * 128, off[i], lds0 + stage * TILE + (DPW * wave + i) * 1024);
  };
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) issue(a, a);
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int stage = 0;
  for (int k = 0; k < steps; ++k) {
    wait_vmcnt<(AHEAD - 1) * DPW>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(stage == 0 ? STAGES - 1 : stage - 1, k + AHEAD);
    acc += *reinterpret_cast<const unsigned*>(smem + stage * TILE + threadIdx.x * 16);
    for (int i = 0; i < spin; ++i) asm volatile("v_mov_b32 %0, %0" : "+v"(acc));  // "compute": 4 cycles per iteration and wave
    stage = stage == STAGES - 1 ? 0 : stage + 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  wait_vmcnt<0>();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = acc; }
}

template <int DPW, int AHEAD, int SWZ = 0>
static int run(const char* src, size_t row_bytes, int steps, int wgs, int spin, unsigned long long* dout) {
  constexpr int TILE = 4 * DPW * 1024;
  const size_t smem = (size_t)(AHEAD + 1) * TILE;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ring_kernel<DPW, AHEAD, SWZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((ring_kernel<DPW, AHEAD, SWZ>), dim3(wgs), dim3(256), smem, 0, src, row_bytes, steps, spin, dout);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((ring_kernel<DPW, AHEAD, SWZ>), dim3(wgs), dim3(256), smem, 0, src, row_bytes, steps, spin, dout);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * wgs);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0; for (int i = 0; i < wgs; ++i) cyc += (double)h[2 * i]; cyc /= wgs;
  printf("swz %d tile %3d KiB ahead %d stages %d (LDS %3zu KiB) wgs %3d spin %4d: %7.0f cycles/step/wg, kernel %6.1f us, aggregate %5.2f TB/s\n", SWZ, TILE / 1024, AHEAD,
         AHEAD + 1, smem / 1024, wgs, spin, cyc / steps, ms * 1e3, (double)wgs * steps * TILE / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  const int rows = 128, K = 4096 * 2;  // bf16 [128][4096]: 8 KiB rows, 1 MiB: the activation of a Llama-3-8B linear at M = 128
  char* src; CK(hipMalloc(&src, (size_t)rows * K + (1 << 20))); CK(hipMemset(src, 1, (size_t)rows * K + (1 << 20)));
  unsigned long long* dout; CK(hipMalloc(&dout, 8192 * 16));
  const int steps = K / 128;  // 64 steps of 128 B per row
  for (int wgs : {224}) {  // round 5: the swizzle question first
    run<4, 2, 0>(src, K, steps, wgs, 0, dout);
    run<4, 2, 1>(src, K, steps, wgs, 0, dout);
    run<4, 2, 2>(src, K, steps, wgs, 0, dout);
    run<4, 2, 3>(src, K, steps, wgs, 0, dout);
    run<4, 2, 4>(src, K, steps, wgs, 0, dout);
    run<4, 4, 0>(src, K, steps, wgs, 0, dout);
    run<4, 4, 1>(src, K, steps, wgs, 0, dout);
    run<8, 2, 0>(src, K, steps, wgs, 0, dout);
    run<8, 2, 1>(src, K, steps, wgs, 0, dout);
  }
  for (int wgs : {56, 112, 224, 448}) {
    for (int spin : {0, 200}) {
      run<4, 2>(src, K, steps, wgs, spin, dout);   // 16 KiB tile (64 rows x 256 B equivalent), 2 ahead
      run<8, 2>(src, K, steps, wgs, spin, dout);   // 32 KiB tile (the int4 kernel's 128 rows x 256 B), 2 ahead, 96 KiB
      run<4, 4>(src, K, steps, wgs, spin, dout);   // 16 KiB, 4 ahead, 80 KiB
      run<2, 2>(src, K, steps, wgs, spin, dout);   // 8 KiB (the MX kernel's 64 rows x 128 B), 2 ahead
      run<2, 5>(src, K, steps, wgs, spin, dout);   // 8 KiB, 5 ahead
    }
  }
  return 0;
}
