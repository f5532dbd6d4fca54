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


// ---- MXFP8 (to_mx, prototype/mx_formats/mx_tensor.py:228-409) --------------------------------------------------------------------
// E8M0 scale exponent of one 32-block from its amax (:255-330; RCEIL :111-129, :161-225) and the reciprocal 2^(127 - e) built from the
// E8M0 byte 254 - e (:132-158).  MODE: AO_MX_SCALE_FLOOR (0) / AO_MX_SCALE_RCEIL (1).
template <int MODE>
__device__ __forceinline__ uint32_t mx_block_exponent(float m, bool finite) {
  uint32_t e;
  if (MODE == 1) {
    // descale = amax * (1/448) in fp32; its value rounded up to a power of two
    const uint32_t bits = f32_to_bits(m * (1.0f / 448.0f));
    const uint32_t be = (bits >> 23) & 0xffu, mant = bits & 0x7fffffu;
    const bool up = (be == 0) ? (mant > 0x400000u) : (mant != 0);
    e = be + (up ? 1u : 0u);
  } else {
    // floor(log2(amax)) - 8, clamped to [-127, 128], biased
    const int ex = (int)((f32_to_bits(m) >> 23) & 0xffu) - 127 - 8;
    e = (uint32_t)(min(max(ex, -127), 128) + 127);
  }
  return finite ? e : 255u;
}
__device__ __forceinline__ float mx_reciprocal(uint32_t e) {
  const uint32_t re = (254u - e) & 0xffu;
  uint32_t rbits = re << 23;
  if (re == 0u) rbits = 0x00400000u;    // 2^-127 as an fp32 subnormal
  if (re == 255u) rbits = 0x7F800001u;  // NaN
  return bits_to_f32(rbits);
}
// The 1 x 32 cast of one block by FOUR ADJACENT LANES (lane & 3 = the block's quarter: 8 bf16 each): block amax across the four lanes,
// the E8M0 exponent (returned in e, the same in all four), 8 e4m3 codes of this lane's quarter.  One definition for the stand-alone cast
// (quant_kernels.hip: mxfp8_quant_kernel) and the cast fused into the grouped GEMM's A-fill (rb8_kernels.hip): the same bits by construction.
template <int MODE>
__device__ __forceinline__ u32x2 mx_cast8(const u32x4& v, uint32_t& e) {
  bool has_nan = false;
  float m = amax8(v, has_nan);
  m = fmaxf(m, __shfl_xor(m, 1));
  m = fmaxf(m, __shfl_xor(m, 2));
  uint32_t nanbits = has_nan ? 1u : 0u;
  nanbits |= __shfl_xor(nanbits, 1);
  nanbits |= __shfl_xor(nanbits, 2);
  e = mx_block_exponent<MODE>(m, (nanbits == 0u) && (m < INFINITY));
  const float r = mx_reciprocal(e);
  float f[8] = {bf16_lo_to_f32(v.x), bf16_hi_to_f32(v.x), bf16_lo_to_f32(v.y), bf16_hi_to_f32(v.y),
                bf16_lo_to_f32(v.z), bf16_hi_to_f32(v.z), bf16_lo_to_f32(v.w), bf16_hi_to_f32(v.w)};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f[j] *= r;
    if (MODE == 0) f[j] = clamp448(f[j]);  // eager saturation (torch < 2.13), :361-373
  }
  return u32x2{cvt4_e4m3(f[0], f[1], f[2], f[3]), cvt4_e4m3(f[4], f[5], f[6], f[7])};
}

}  // namespace ao
