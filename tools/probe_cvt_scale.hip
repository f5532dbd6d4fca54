// Semantics probe: does v_cvt_scalef32_pk_{f32,bf16}_fp8 multiply by an ARBITRARY fp32 scale
// (and round the bf16 result RNE), or does it only use the scale's exponent?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* scales, int ns, float* out_f32, uint32_t* out_bf16) {
  const int code = threadIdx.x;  // 0..255
  for (int i = 0; i < ns; ++i) {
    const float sc = scales[i];
    const uint32_t src = (uint32_t)code | ((uint32_t)code << 8);
    f32x2 r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(src, sc, false);
    out_f32[i * 256 + code] = r.x;
    bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, sc, false);
    out_bf16[i * 256 + code] = __builtin_bit_cast(uint32_t, b) & 0xffffu;
  }
}
static float e4m3(int c) {
  int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
  float v;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) v = m * std::ldexp(1.0f, -9); else v = (1.0f + m / 8.0f) * std::ldexp(1.0f, e - 7);
  return s ? -v : v;
}
static uint32_t bf16_rne(float f) { uint32_t u; memcpy(&u, &f, 4); if (std::isnan(f)) return 0x7fc0; u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
int main() {
  std::vector<float> sc = {1.0f, 512.0f, 512.0f * 0.00390625f, 1.5f, 3.0f, 0.0123291015625f * 512.f, 1.3359375f * 512.f, 0.00201416015625f * 512.f, 1.9921875f, 1.0078125f * 512.f};
  float* ds; hipMalloc(&ds, sc.size() * 4); hipMemcpy(ds, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
  float* of; uint32_t* ob; hipMalloc(&of, sc.size() * 256 * 4); hipMalloc(&ob, sc.size() * 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, ds, (int)sc.size(), of, ob); hipDeviceSynchronize();
  std::vector<float> hf(sc.size() * 256); std::vector<uint32_t> hb(sc.size() * 256);
  hipMemcpy(hf.data(), of, hf.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), ob, hb.size() * 4, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < sc.size(); ++i) {
    int bad_f = 0, bad_b = 0, bad_f_exp = 0;
    int ex; std::frexp(sc[i], &ex); const float pow2 = std::ldexp(1.0f, ex - 1);
    for (int c = 0; c < 256; ++c) {
      const float v = e4m3(c); if (std::isnan(v)) continue;
      const float want = v * sc[i];  // exact in fp32 (4-bit x 24-bit)
      if (hf[i * 256 + c] != want) ++bad_f;
      if (hf[i * 256 + c] != v * pow2) ++bad_f_exp;
      if (hb[i * 256 + c] != bf16_rne(want)) ++bad_b;
    }
    printf("scale %-14.9g: f32 out != v*scale: %3d (!= v*2^floor(log2 scale): %3d); bf16 out != rne(v*scale): %3d   e.g. code 0x0b(=11*2^-9): f32 %g bf16 0x%04x want %g 0x%04x\n",
           sc[i], bad_f, bad_f_exp, bad_b, hf[i * 256 + 0x0b], hb[i * 256 + 0x0b], e4m3(0x0b) * sc[i], bf16_rne(e4m3(0x0b) * sc[i]));
  }
  return 0;
}
