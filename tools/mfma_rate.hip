// Profiling aid: issue rate of the 1-byte MFMAs on gfx950 (cycles per instruction from s_memtime, chip throughput from wall time).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate && tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(512) void k(unsigned long long* o, int iters, int seed) {
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i); b[i] = seed * (threadIdx.x * 3 + i); }
  f32x4 c4[NACC];
  f32x16 c16[NACC > 4 ? 4 : NACC];
  for (int i = 0; i < NACC; ++i) c4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < (NACC > 4 ? 4 : NACC); ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (KIND == 0) {
        i32x4 c = __builtin_bit_cast(i32x4, c4[i]);
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(i32x4{a[0], a[1], a[2], a[3]}, i32x4{b[0], b[1], b[2], b[3]}, c, 0, 0, 0);
        c4[i] = __builtin_bit_cast(f32x4, c);
      } else if constexpr (KIND == 1) {
        c4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c4[i], 0, 0, 0, 127, 0, 127);
      } else if constexpr (KIND == 2) {
        if (i < 4) c16[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c16[i], 0, 0, 0, 127, 0, 127);
      } else if constexpr (KIND == 3) {
        c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, i32x4{a[0], a[1], a[2], a[3]}), __builtin_bit_cast(bf16x8, i32x4{b[0], b[1], b[2], b[3]}), c4[i], 0, 0, 0);
      } else {  // non-scaled fp8 16x16x32
        c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(((long)a[1] << 32) | (unsigned)a[0], ((long)b[1] << 32) | (unsigned)b[0], c4[i], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += c4[i][0] + c4[i][3];
  for (int i = 0; i < (NACC > 4 ? 4 : NACC); ++i) s += c16[i][0] + c16[i][15];
  if ((threadIdx.x & 63) == 0) o[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  if (s == 12345.678f) o[0] = 1;
}

template <int KIND, int NACC>
void run(const char* name, double flop_per_inst, int n_inst_per_iter) {
  unsigned long long* d;
  hipMalloc(&d, 8 * 1024 * 8);
  const int iters = 20000;
  for (int waves : {4, 8}) {  // per CU: 1 or 2 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(256), dim3(64 * waves), 0, 0, d, 100, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(256), dim3(64 * waves), 0, 0, d, iters, 3);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double insts = (double)iters * n_inst_per_iter;
    printf("%-28s %d waves/CU: %.1f ticks per MFMA per wave, wall %.3f ms -> %.0f TFLOP/s chip (%.0f MHz effective by ticks)\n", name, waves,
           (double)h[0] / insts, ms, insts * flop_per_inst * 256.0 * waves / (ms * 1e-3) / 1e12, (double)h[0] / (ms * 1e3));
  }
  hipFree(d);
}

int main() {
  run<0, 8>("i8 16x16x64", 2.0 * 16 * 16 * 64, 8);
  run<1, 8>("f8f6f4 16x16x128 (e4m3)", 2.0 * 16 * 16 * 128, 8);
  run<2, 4>("f8f6f4 32x32x64 (e4m3)", 2.0 * 32 * 32 * 64, 4);
  run<3, 8>("bf16 16x16x32", 2.0 * 16 * 16 * 32, 8);
  run<4, 8>("fp8 16x16x32 (non-scaled)", 2.0 * 16 * 16 * 32, 8);
  return 0;
}
