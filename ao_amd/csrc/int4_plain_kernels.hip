// Int4Tensor (PLAIN packing) weight preparation for gfx950: the default format of the reference's Int4WeightOnlyConfig and of
// Float8DynamicActivationInt4WeightConfig (quantize_/workflows/int4/int4_tensor.py:130-186).
//
// The reference delegates the arithmetic to the un-vendored `mslk` package (mslk.quantize.shuffle.int4_row_quantize_zp /
// int4_row_quantize / pack_int4, pinned mslk==1.3.0 in .github/workflows/1xH100_tests.yml:33).  Its published algorithm is
// restated inside the reference itself -- torchao/quantization/qat/fake_quantizer.py:148-190 ("simulates the numerics of
// mslk.quantize.shuffle.int4_row_quantize[_zp]") and torchao/prototype/gptq/api.py:167-221 -- and that is what this kernel
// replays, in fp32 like them:
//   asymmetric (bf16 activations):  scale = max(max - min, 1e-6) / 15;  zero = min + 8 scale;
//                                   q = clamp(rint((w - min) / scale), 0, 15) - 8
//   symmetric  (fp8 activations):   scale = max(max|w| / 8, 1e-6);  zero = 0;  q = clamp(rint(w / scale), -8, 7)
//   qdata[n][k/2] = (q[n][k] & 0xF) | (q[n][k+1] << 4)   (pack_int4: even k in the LOW nibble)
//   scale, zero: [K/g][N] in the weight dtype (bf16)
// The MATMUL needs no kernel of its own: (q, scale, zero) of this format and of the tinygemm tile-packed format describe the
// same dequantised weight  bf16(bf16(q * scale) + zero)  (zero = the value of q = 0 signed = 8 unsigned in both), so the
// host side re-lays a PLAIN weight into the tile-packed layout once (ao_amd/quantization/int4_plain_tensor.py) and the
// existing _weight_int4pack_mm kernels serve it.
#include "common.h"

namespace ao {
namespace {

// One wave = 4 rows x 128 k per block; lane (row = l >> 4, chunk = l & 15) owns 8 consecutive k (one 16-byte load).
template <int G, bool SYM>
__global__ __launch_bounds__(64) void int4_plain_quantize_kernel(const uint16_t* __restrict__ w, uint8_t* __restrict__ qdata,
                                                                 uint16_t* __restrict__ scale, uint16_t* __restrict__ zero, int64_t N,
                                                                 int64_t K) {
  constexpr int KB = (G > 128) ? (G / 128) : 1;    // 128-k blocks per group
  constexpr int LANES = (G >= 128) ? 16 : (G / 8);  // lanes of a row that share a group inside one block
  const int lane = threadIdx.x;
  const int64_t spans = K / (128 * KB);
  const int64_t row = ((int64_t)blockIdx.x / spans) * 4 + (lane >> 4);
  const int64_t k0 = ((int64_t)blockIdx.x % spans) * (128 * KB) + (lane & 15) * 8;
  const bool live = row < N;
  float v[KB][8];
  float mx = -INFINITY, mn = INFINITY;
#pragma unroll
  for (int b = 0; b < KB; ++b) {
    u32x4 raw = {0u, 0u, 0u, 0u};
    if (live) raw = *reinterpret_cast<const u32x4*>(w + row * K + k0 + b * 128);
    const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[b][2 * i] = bf16_lo_to_f32(r[i]);
      v[b][2 * i + 1] = bf16_hi_to_f32(r[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = SYM ? fabsf(v[b][i]) : v[b][i];
      mx = fmaxf(mx, a);
      mn = fminf(mn, a);
    }
  }
#pragma unroll
  for (int off = 1; off < LANES; off <<= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, off));
    mn = fminf(mn, __shfl_xor(mn, off));
  }
  float s, z, base;
  if (SYM) {
    s = fmaxf(mx / 8.0f, 1e-6f);
    z = 0.f;
    base = 0.f;
  } else {
    s = fmaxf(mx - mn, 1e-6f) / 15.0f;
    z = mn + s * 8.0f;
    base = mn;
  }
#pragma unroll
  for (int b = 0; b < KB; ++b) {
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float q = rintf((v[b][i] - base) / s);  // IEEE division, like torch
      q = SYM ? fminf(fmaxf(q, -8.f), 7.f) : (fminf(fmaxf(q, 0.f), 15.f) - 8.f);
      word |= ((uint32_t)(int)q & 0xFu) << (4 * i);  // even k -> low nibble of its byte
    }
    if (live) *reinterpret_cast<uint32_t*>(qdata + row * (K / 2) + (k0 + b * 128) / 2) = word;
  }
  if (live && (lane & (LANES - 1)) == 0) {
    const int64_t g = k0 / G;
    scale[g * N + row] = f32_to_bf16_bits(s);
    zero[g * N + row] = f32_to_bf16_bits(z);
  }
}

template <bool SYM>
int launch_plain_quantize(const uint16_t* w, uint8_t* qdata, uint16_t* scale, uint16_t* zero, int64_t N, int64_t K, int g, hipStream_t s) {
  const int64_t kb = g > 128 ? g / 128 : 1;
  const int64_t blocks = ((N + 3) / 4) * (K / (128 * kb));
  AO_REQUIRE(blocks < (1ll << 31), "ao_int4_plain_quantize: tensor too large for one launch");
  dim3 grid((unsigned)blocks), block(64);
  switch (g) {
    case 32: ao::launch(int4_plain_quantize_kernel<32, SYM>, grid, block, 0, s, w, qdata, scale, zero, N, K); break;
    case 64: ao::launch(int4_plain_quantize_kernel<64, SYM>, grid, block, 0, s, w, qdata, scale, zero, N, K); break;
    case 128: ao::launch(int4_plain_quantize_kernel<128, SYM>, grid, block, 0, s, w, qdata, scale, zero, N, K); break;
    default: ao::launch(int4_plain_quantize_kernel<256, SYM>, grid, block, 0, s, w, qdata, scale, zero, N, K); break;
  }
  AO_LAUNCH_CHECK("int4_plain_quantize_kernel launch");
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int ao_int4_plain_quantize(const uint16_t* w, uint8_t* qdata, uint16_t* scale, uint16_t* zero_point, int64_t N, int64_t K,
                                      int group_size, int symmetric, void* stream) {
  AO_REQUIRE(N > 0 && K > 0, "ao_int4_plain_quantize: bad shape N=%lld K=%lld", (long long)N, (long long)K);
  AO_REQUIRE(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256,
             "ao_int4_plain_quantize: group_size must be one of 32, 64, 128, 256, got %d", group_size);
  AO_REQUIRE(K % 128 == 0 && K % group_size == 0, "ao_int4_plain_quantize: K=%lld must be a multiple of 128 and of group_size=%d", (long long)K,
             group_size);
  AO_REQUIRE_PTR(w);
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE_PTR(scale);
  AO_REQUIRE_PTR(zero_point);
  return symmetric ? launch_plain_quantize<true>(w, qdata, scale, zero_point, N, K, group_size, (hipStream_t)stream)
                   : launch_plain_quantize<false>(w, qdata, scale, zero_point, N, K, group_size, (hipStream_t)stream);
}
