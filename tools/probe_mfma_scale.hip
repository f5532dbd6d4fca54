// Probe: lane -> (row, k-block) mapping of the scale operands of
// v_mfma_scale_f32_16x16x128_f8f6f4 (and 32x32x64) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const int* sa, const int* sb, const int* adata, float* out) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = adata[l * 8 + i]; b[i] = 0x38383838; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void k32(const int* sa, const int* sb, float* out) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
  int *dsa, *dsb, *dad; float* dout;
  hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dad, 64 * 32); hipMalloc(&dout, 64 * 16 * 4);
  std::vector<int> sa(64, 127), sb(64, 127), ad(64 * 8, 0x38383838);
  std::vector<float> o(64 * 16);
  auto run16 = [&]() {
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dad, ad.data(), 64 * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dsa, dsb, dad, dout); hipDeviceSynchronize();
    hipMemcpy(o.data(), dout, 64 * 4 * 4, hipMemcpyDeviceToHost);
  };
  run16();
  printf("16x16x128 baseline D[0][0]=%g (expect 128)\n", o[0]);
  // D layout: lane l, reg r -> row (l>>4)*4 + r, col l&15
  for (int L : {0, 1, 15, 16, 17, 32, 48, 63}) {
    sa.assign(64, 127); sa[L] = 128; run16();
    printf("scale_a lane %2d doubled -> rows changed:", L);
    for (int row = 0; row < 16; ++row) { float v = o[((row >> 2) * 16 + 0) * 4 + (row & 3)]; if (v != 128) printf(" row%d=%g", row, v); }
    printf("\n");
  }
  sa.assign(64, 127);
  for (int L : {0, 1, 16, 33, 63}) {
    sb.assign(64, 127); sb[L] = 128; run16();
    printf("scale_b lane %2d doubled -> cols changed:", L);
    for (int col = 0; col < 16; ++col) { float v = o[col * 4 + 0]; if (v != 128) printf(" col%d=%g", col, v); }
    printf("\n");
  }
  sb.assign(64, 127);
  // which k-block does lane-group g's scale apply to?  zero out A data of one lane (row 0) -> its 32 k vanish.
  // then double scale of lane L' in same row: if row0 total changes by +32 the scale maps to a live block.
  for (int g = 0; g < 4; ++g) {
    ad.assign(64 * 8, 0x38383838);
    for (int i = 0; i < 8; ++i) ad[(g * 16 + 0) * 8 + i] = 0;  // kill row 0's data held by lane 16g
    for (int g2 = 0; g2 < 4; ++g2) {
      sa.assign(64, 127); sa[g2 * 16] = 128; run16();
      printf("  data of lane %2d zeroed, scale of lane %2d doubled: D[0][0]=%g\n", g * 16, g2 * 16, o[0]);
    }
  }
  ad.assign(64 * 8, 0x38383838); sa.assign(64, 127);
  // byte position: scale in byte 1 with opsel 0 should have no effect if only byte 0 is read
  sa[0] = 127 | (200 << 8); run16(); printf("garbage in byte1 of lane0 scale_a: D[0][0]=%g (128 => only byte0 read)\n", o[0]);
  // 32x32x64
  std::vector<int> s32(64, 127);
  hipMemcpy(dsb, s32.data(), 256, hipMemcpyHostToDevice);
  for (int L : {0, 1, 31, 32, 33, 63}) {
    s32.assign(64, 127); s32[L] = 128; hipMemcpy(dsa, s32.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dsa, dsb, dout); hipDeviceSynchronize();
    hipMemcpy(o.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
    printf("32x32x64 scale_a lane %2d doubled -> rows changed (col 0):", L);
    for (int row = 0; row < 32; ++row) { int r = (row & 3) + 4 * ((row >> 3)); int lane = ((row >> 2) & 1) * 32; float v = o[lane * 16 + r]; if (v != 64) printf(" row%d=%g", row, v); }
    printf("\n");
  }
  return 0;
}
