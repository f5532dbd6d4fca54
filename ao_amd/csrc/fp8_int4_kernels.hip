// fp8-activation x int4-weight linear for gfx950.
//
// Replaces mslk.f8i4bf16_rowwise as called by Int4Tensor's F.linear with activation_dtype = float8_e4m3fn
// (torchao/quantization/quantize_/workflows/int4/int4_tensor.py:213-229; config Float8DynamicActivationInt4WeightConfig,
// quant_api.py:630-699): per-row dynamic e4m3 activations (x = xq * x_scale[m]), groupwise int4 weights
// (w[n][k] = (q[n][k] - 8) * s[k/g][n] + z[k/g][n]; the reference's symmetric flavour has z = 0), bf16 out:
//     y[m][n] = bf16( x_scale[m] * sum_g ( s[g][n] * sum_{k in g} xq[m][k] (q[n][k] - 8)  +  z[g][n] * sum_{k in g} xq[m][k] ) + bias[n] )
// The scale is applied to the fp32 GROUP SUMS, never to single weights: there is no per-weight rounding to replay (unlike the bf16
// tinygemm path), so the codes go to the matrix pipe as they are:
//   * an offset-8 nibble q (0..15) in a byte IS the e4m3 number q * 2^-9 (subnormals and the first binade are linear), so the B
//     operand of v_mfma_f32_16x16x32_fp8_fp8 is two mask operations away from the packed word; products xq * q * 2^-9 are exact in fp32;
//   * the "- 8" and the zero-point need sum_{k in g} xq[m][k]: one more MFMA of the same A fragment against a B operand of ones
//     returns it in the accumulator's own register layout;
//   * per group and 16 x 16 tile: acc += (512 s) * P_q + (z - 8 s) * P_x   (two packed fp32 FMAs per accumulator register).
// Weights are consumed in the tinygemm tile order (Int4Tensor.tile_packed(): [N/16][K/128][64 lanes][16 B], offset-8 codes), the
// layout of int4_mm_kernel: one workgroup per 16-wide n-tile and 16-row slab, waves split K, packed blocks straight into VGPRs
// through a register ring, x slice staged per wave in LDS (A operands: word j of lane (m, kq) = x[m][32 j + 4 kq ..+3] and
// x[m][32 j + 16 + 4 kq ..+3], byte-interleaved with two v_perm to follow the nibble order of the masks).
// ~50 instructions per 1 KiB block instead of ~100 (exact bf16 dequant): at M = 1 this path is bound by HBM, not by VALU issue.
#include <utility>

#include "common.h"

namespace ao {
namespace {

typedef long i64_t;

template <int G, int DEPTH>
__global__ __launch_bounds__(512) void fp8_int4_mm_kernel(const uint8_t* __restrict__ xq, const float* __restrict__ x_scale,
                                                          const u32x4* __restrict__ qdata, const uint32_t* __restrict__ sz,
                                                          const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int M, int N, int K) {
  constexpr int NG = (G >= 128) ? 1 : (128 / G);  // groups per 128-k block
  constexpr int ROWSTRIDE = 128 + 16;             // bytes per staged x row ([kq][j] pairs of dwords), padded vs bank conflicts
  constexpr int SLAB = 16 * ROWSTRIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int ntile = blockIdx.x;
  const int m0 = blockIdx.y * 16;
  const int rows = min(16, M - m0);
  const int kblocks = K >> 7;
  const int kb0 = (kblocks * wave) / nwaves;
  const int kb1 = (kblocks * (wave + 1)) / nwaves;
  char* slab = smem + wave * SLAB;
  float* red = reinterpret_cast<float*>(smem + nwaves * SLAB);

  const int n = ntile * 16 + (lane & 15);
  const int kq = lane >> 4;
  const u32x4* wp = qdata + (size_t)ntile * kblocks * 64 + lane;
  // x slice of a block: lane (r = lane >> 2, j = lane & 3) loads the 32 bytes x[m0 + r][kb * 128 + 32 j ..] (rows past M: the last row)
  const uint8_t* xp = xq + (size_t)(m0 + min(lane >> 2, rows - 1)) * K + (lane & 3) * 32;
  // ... and stores the dword pairs (d[kq'], d[4 + kq']) at [r][kq'][j]; lane (m, kq) reads its four pairs (j = 0..3) as 32 contiguous bytes
  char* st_base = slab + (lane >> 2) * ROWSTRIDE + (lane & 3) * 8;
  const char* a_base = slab + (lane & 15) * ROWSTRIDE + kq * 32;

  struct Stage {
    u32x4 w;
    uint32_t sz[NG];
    u32x4 x0, x1;
  };
  Stage st[DEPTH];
  auto issue = [&](Stage& s, int kb) {
    s.w = __builtin_nontemporal_load(wp + (size_t)kb * 64);
    const int kg0 = (G >= 128) ? ((kb * 128) / G) : (kb * NG);
#pragma unroll
    for (int i = 0; i < NG; ++i) s.sz[i] = sz[(size_t)(kg0 + i) * N + n];
    const u32x4* xs = reinterpret_cast<const u32x4*>(xp + (size_t)kb * 128);
    s.x0 = xs[0];
    s.x1 = xs[1];
  };

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const i64_t ones = 0x3838383838383838L;  // eight e4m3 1.0
  auto consume = [&](const Stage& s) {
    // stage x: d0..d3 = tile 2j (kq 0..3), d4..d7 = tile 2j + 1
    const uint32_t d[8] = {s.x0.x, s.x0.y, s.x0.z, s.x0.w, s.x1.x, s.x1.y, s.x1.z, s.x1.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2*>(st_base + q * 32) = u32x2{d[q], d[4 + q]};
    const u32x4 xa = *reinterpret_cast<const u32x4*>(a_base);       // (A0, B0, A1, B1)
    const u32x4 xb = *reinterpret_cast<const u32x4*>(a_base + 16);  // (A2, B2, A3, B3)
    const uint32_t xw[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    const uint32_t wds[4] = {s.w.x, s.w.y, s.w.z, s.w.w};
    f32x4 pq = {0.f, 0.f, 0.f, 0.f}, px = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // B operand: bytes (v0, v4, v1, v5 | v2, v6, v3, v7) = k (t0, t1, t0 + 1, t1 + 1 | t0 + 2, t1 + 2, t0 + 3, t1 + 3)
      const uint32_t lo = wds[j] & 0x0F0F0F0Fu, hi = (wds[j] >> 4) & 0x0F0F0F0Fu;
      // A operand in the same order: interleave the bytes of A_j = x[.., t0 ..+3] and B_j = x[.., t1 ..+3]
      const uint32_t alo = __builtin_amdgcn_perm(xw[2 * j + 1], xw[2 * j], 0x05010400u);
      const uint32_t ahi = __builtin_amdgcn_perm(xw[2 * j + 1], xw[2 * j], 0x07030602u);
      const i64_t a = (i64_t)(((unsigned long)ahi << 32) | alo), b = (i64_t)(((unsigned long)hi << 32) | lo);
      pq = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, pq, 0, 0, 0);
      px = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, ones, px, 0, 0, 0);
      if ((j + 1) * 32 % G == 0 || j == 3) {  // a group ends here (G >= 128: once per block)
        const int gi = (G >= 128) ? 0 : ((j * 32) / G);
        const float sc = bf16_lo_to_f32(s.sz[gi]), zp = bf16_hi_to_f32(s.sz[gi]);
        const float c0 = 512.0f * sc, c1 = zp - 8.0f * sc;
        acc.x = fmaf(c0, pq.x, fmaf(c1, px.x, acc.x)); acc.y = fmaf(c0, pq.y, fmaf(c1, px.y, acc.y));
        acc.z = fmaf(c0, pq.z, fmaf(c1, px.z, acc.z)); acc.w = fmaf(c0, pq.w, fmaf(c1, px.w, acc.w));
        pq = f32x4{0.f, 0.f, 0.f, 0.f}; px = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  const int kb_last = max(kb1 - 1, kb0);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(st[d], min(kb0 + d, kb_last));
  auto for_slots = [&](auto&& f) {
    [&]<int... D>(std::integer_sequence<int, D...>) { (f(std::integral_constant<int, D>{}), ...); }(std::make_integer_sequence<int, DEPTH>{});
  };
  int kb = kb0;
  for (; kb + 2 * DEPTH <= kb1; kb += DEPTH) {
    for_slots([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      consume(st[d]);
      issue(st[d], kb + d + DEPTH);
    });
  }
  for_slots([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    if (kb + d < kb1) {
      consume(st[d]);
      if (kb + d + DEPTH < kb1) issue(st[d], kb + d + DEPTH);
    }
  });
  kb += DEPTH;
  for_slots([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    if (kb + d < kb1) consume(st[d]);
  });

  // cross-wave reduction: red[wave][row][col]; D layout: lane (col = lane & 15, group kq) holds rows 4 kq + {0..3}
  {
    float* r = red + wave * 256 + (kq * 4) * 16 + (lane & 15);
    r[0] = acc.x; r[16] = acc.y; r[32] = acc.z; r[48] = acc.w;
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < 256) {
    const int row = tid >> 4, col = tid & 15;
    if (row < rows) {
      float sum = 0.f;
      for (int w = 0; w < nwaves; ++w) sum += red[w * 256 + tid];
      float v = sum * x_scale[m0 + row];
      if (bias != nullptr) v += bf16_lo_to_f32(bias[ntile * 16 + col]);
      y[(size_t)(m0 + row) * N + ntile * 16 + col] = f32_to_bf16_bits(v);
    }
  }
}

template <int G>
int launch_fp8_int4(const uint8_t* xq, const float* x_scale, const int32_t* qdata, const uint16_t* sz, const uint16_t* bias, uint16_t* y,
                    int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  const int kblocks = (int)(K >> 7);
  int wpb = (kblocks >= 16) ? 8 : 4;
  if (wpb > kblocks) wpb = kblocks < 4 ? 4 : kblocks;
  const size_t smem = (size_t)wpb * (16 * (128 + 16) + 1024);
  dim3 grid((unsigned)(N >> 4), (unsigned)((M + 15) / 16)), block(wpb * 64);
  ao::launch(fp8_int4_mm_kernel<G, 4>, grid, block, smem, stream, xq, x_scale, reinterpret_cast<const u32x4*>(qdata),
             reinterpret_cast<const uint32_t*>(sz), bias, y, (int)M, (int)N, (int)K);
  AO_LAUNCH_CHECK("fp8_int4_mm_kernel launch");
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int ao_fp8_int4_linear(const uint8_t* xq, const float* x_scale, const int32_t* qdata, const uint16_t* scale_and_zero,
                                  const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K, int group_size, void* stream) {
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE_PTR(scale_and_zero);
  AO_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0, "ao_fp8_int4_linear: N=%lld must be a multiple of 16 and K=%lld of 128", (long long)N,
             (long long)K);
  AO_REQUIRE(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256, "ao_fp8_int4_linear: group_size must be 32, 64, 128 or 256, got %d",
             group_size);
  AO_REQUIRE(K % group_size == 0, "ao_fp8_int4_linear: K=%lld not divisible by group_size=%d", (long long)K, group_size);
  AO_REQUIRE(M >= 0 && M < (1ll << 20) && N < (1ll << 31) && K < (1ll << 31), "ao_fp8_int4_linear: bad M=%lld", (long long)M);
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(xq);
  AO_REQUIRE_PTR(x_scale);
  AO_REQUIRE_PTR(y);
  hipStream_t s = (hipStream_t)stream;
  switch (group_size) {
    case 32: return launch_fp8_int4<32>(xq, x_scale, qdata, scale_and_zero, bias, y, M, N, K, s);
    case 64: return launch_fp8_int4<64>(xq, x_scale, qdata, scale_and_zero, bias, y, M, N, K, s);
    case 128: return launch_fp8_int4<128>(xq, x_scale, qdata, scale_and_zero, bias, y, M, N, K, s);
    default: return launch_fp8_int4<256>(xq, x_scale, qdata, scale_and_zero, bias, y, M, N, K, s);
  }
}
