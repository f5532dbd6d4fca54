// Profiling aid: does a wave that issues 1-byte MFMAs slow down a ds_read_b128 wave on the same SIMD, and by how much per MFMA kind?
// Block = 8 waves (2 per SIMD): waves 0-3 multiply, waves 4-7 read LDS (conflict-free 16 B per lane) -- the two roles of gemm8_p8's phases.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue.hip -o tools/bin/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* o, int iters, int seed) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<int*>(lds)[i] = i * seed;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float s = 0;
  if (wave < 4) {
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i); b[i] = seed * (threadIdx.x * 3 + i); }
    f32x4 c4[8];
    f32x16 c16[4];
    for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
      if constexpr (KIND == 0) {  // 16 x i8 16x16x64 = 256 cycles
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            i32x4 c = __builtin_bit_cast(i32x4, c4[i]);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(i32x4{a[0], a[1], a[2], a[3]}, i32x4{b[0], b[1], b[2], b[3]}, c, 0, 0, 0);
            c4[i] = __builtin_bit_cast(f32x4, c);
          }
      } else if constexpr (KIND == 1) {  // 8 x f8f6f4 16x16x128 = 256 cycles
#pragma unroll
        for (int i = 0; i < 8; ++i) c4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c4[i], 0, 0, 0, 127, 0, 127);
      } else if constexpr (KIND == 2) {  // 4 x f8f6f4 32x32x64 = 256 cycles
#pragma unroll
        for (int i = 0; i < 4; ++i) c16[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c16[i], 0, 0, 0, 127, 0, 127);
      } else {  // no MFMA: the readers alone
      }
    }
    for (int i = 0; i < 8; ++i) s += c4[i][0];
    for (int i = 0; i < 4; ++i) s += c16[i][0];
  } else {
    u32x4 acc = {0, 0, 0, 0};
    const char* base = lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {  // 16 reads per iteration, like a gemm8_p8 load phase
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((it + j) & 63) * 1024);
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
      }
    }
    s = (float)(acc.x + acc.y + acc.z + acc.w);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) o[blockIdx.x * 8 + wave] = t1 - t0;
  if (s == 12345.678f) o[0] = 1;
}

template <int KIND>
void run(const char* name) {
  unsigned long long* d;
  (void)hipMalloc(&d, 256 * 8 * 8);
  const int iters = 4000;
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, d, 10, 3);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, d, iters, 3);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%-26s wall %.3f ms | MFMA wave: %.1f ticks per 256-cycle group | LDS wave: %.1f ticks per 16 ds_read_b128\n", name, ms, (double)h[0] / iters, (double)h[4] / iters);
  (void)hipFree(d);
}

int main() {
  run<3>("readers alone");
  run<0>("i8 16x16x64 x16");
  run<1>("f8f6f4 16x16x128 x8");
  run<2>("f8f6f4 32x32x64 x4");
  return 0;
}
