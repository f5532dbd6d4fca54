// Per-row activation quantisation arithmetic shared by the stand-alone casts (quant_kernels.hip) and the fused
// dynamic-quant linears (dyn8_kernels.hip): the reference's op sequence, element for element.
//   int8 : scale = f32(max(bf16(amax / 127.5), bf16(f32_eps)));  q = clamp(rint(x * (1/scale)), -128, 127)
//          (int8_tensor.py:191-230, quant_primitives.py:1534-1583, :463-485)
//   fp8  : scale = f32(bf16(amax / 448));  q = e4m3_rne(clamp(f32(x) / scale, -448, 448))
//          (float8_tensor.py:167-253, quant_primitives.py:2192-2212, 2271-2287)
#pragma once
#include "common.h"

namespace ao {

// NaN-propagating max like torch.amax: fmaxf drops NaN, so track it separately
__device__ __forceinline__ float amax8(const u32x4& v, bool& has_nan) {
  float m = 0.f;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = fabsf(bf16_lo_to_f32(w[i])), b = fabsf(bf16_hi_to_f32(w[i]));
    has_nan |= (a != a) | (b != b);
    m = fmaxf(m, fmaxf(a, b));
  }
  return m;
}

__device__ __forceinline__ float int8_row_scale(float amax) {
  const float s = round_bf16(amax / 127.5f);
  return fmaxf(s, 1.1920928955078125e-07f);  // fp32 eps, exactly representable in bf16
}
// 8 bf16 -> 8 int8 (two dwords); inv = 1 / scale
__device__ __forceinline__ u32x2 int8_quant8(const u32x4& v, float inv) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t out[2] = {0u, 0u};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = fminf(fmaxf(rintf(bf16_lo_to_f32(w[j]) * inv), -128.f), 127.f);
    const float b = fminf(fmaxf(rintf(bf16_hi_to_f32(w[j]) * inv), -128.f), 127.f);
    const uint32_t pa = (uint32_t)(int)a & 0xffu, pb = (uint32_t)(int)b & 0xffu;
    out[j >> 1] |= (pa | (pb << 8)) << ((j & 1) * 16);
  }
  return u32x2{out[0], out[1]};
}

__device__ __forceinline__ float fp8_row_scale(float amax) { return round_bf16(amax / 448.0f); }
__device__ __forceinline__ uint32_t cvt4_e4m3(float a, float b, float c, float d) {
  uint32_t r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return r;
}
__device__ __forceinline__ float clamp448(float v) {
  // torch.clamp propagates NaN; fminf/fmaxf would not
  return (v != v) ? v : fminf(fmaxf(v, -448.f), 448.f);
}
// 8 bf16 -> 8 e4m3 (two dwords).  The reference divides (tensor_fp32 / scale).  Here: one IEEE reciprocal per call and a
// residual-corrected product per element, q0 = x r;  q = fma(fma(-q0, s, x), r, q0), which gives the SAME e4m3 codes: when x / s
// is exactly representable (the only way it can sit on an e4m3 rounding boundary or on the 448 clamp: x and s are bf16-valued)
// the correction recovers it exactly; otherwise it is within an ulp of the correctly rounded quotient and at least 2^-13
// (relative) away from any boundary.  tests/test_oracle_variants.py checks every bf16 x against scales over 200 binades.  Scales
// whose reciprocal would overflow or go denormal take the division.
__device__ __forceinline__ u32x2 fp8_quant8(const u32x4& v, float s) {
  float f[8] = {bf16_lo_to_f32(v.x), bf16_hi_to_f32(v.x), bf16_lo_to_f32(v.y), bf16_hi_to_f32(v.y),
                bf16_lo_to_f32(v.z), bf16_hi_to_f32(v.z), bf16_lo_to_f32(v.w), bf16_hi_to_f32(v.w)};
  if (s > 0x1p-100f && s < 0x1p100f) {  // uniform per row
    const float r = 1.0f / s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float q0 = f[j] * r;
      const float q = __builtin_fmaf(__builtin_fmaf(-q0, s, f[j]), r, q0);
      f[j] = clamp448((fabsf(q0) < INFINITY && q0 != 0.0f) ? q : q0);  // inf / NaN / signed zero pass through like the division
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = clamp448(f[j] / s);
  }
  return u32x2{cvt4_e4m3(f[0], f[1], f[2], f[3]), cvt4_e4m3(f[4], f[5], f[6], f[7])};
}

}  // namespace ao
