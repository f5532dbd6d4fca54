// Dynamic-activation 8-bit linears at decode sizes with the activation cast FUSED into the matmul:
//
//   int8:  xq, sx = per-row symmetric int8 of x;  y = bf16(bf16(i32(xq . wq^T) * sx[m]) * sw[n] + bias[n])
//          (Int8Tensor F.linear with act_quant_kwargs: int8_tensor.py:176-248 + :305-359)
//   fp8 :  xq, sx = per-row e4m3 of x;            y = bf16((xq . wq^T) * sx[m] * sw[n] + bias[n])
//          (Float8Tensor F.linear with act_quant_kwargs: float8_tensor.py:167-253 + float8/inference.py:104-123)
//
// One launch instead of two (the stand-alone cast costs 2.7-5 us per linear at M = 1, as much as a small projection).
// Structure = the per-tile weight-streaming kernel (stream8_kernels.hip): one workgroup per 16-wide n-tile, waves split
// K, weights straight into VGPRs.  Every workgroup first casts the whole activation itself -- M x K bf16 from L2, at
// most 64 KiB of codes -- into LDS with exactly the arithmetic of the stand-alone kernels (quant_math.h), then feeds
// its MFMAs from there; rows are K + 16 bytes apart so that the 16-byte operand reads of different rows hit different
// banks.  Same bits as cast + matmul (int8 bit-exact end to end).
#include "common.h"
#include "quant_math.h"

#include <algorithm>

namespace ao {
// dec8_kernels.hip (round 4): the same linear with the weights as full lines in a register ring and the cast under their flight
bool dec8_takes(int64_t M, int64_t N, int64_t K);
int dec8_dynamic(bool int8, const uint16_t* x, const void* wq, const float* w_scale, const uint16_t* bias, uint16_t* y, int64_t M, int64_t N,
                 int64_t K, hipStream_t stream);
// mid8_kernels.hip (round 4): 16 < M <= 256 with few output tiles -- the per-row cast shared out among the workgroups of the same launch
bool mid8_takes_fused(int64_t M, int64_t N, int64_t K);
int mid8_dynamic(bool int8, const uint16_t* x, const void* b, const float* scale_b, const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K,
                 hipStream_t stream);
namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct Dyn8Args {
  const uint16_t* x;       // [M][K] bf16
  const uint8_t* b;        // [N][K] int8 / e4m3
  const float* col_scale;  // [N]
  const uint16_t* bias;    // [N] bf16 or null
  uint16_t* out;           // [M][N] bf16
  int M, N, K;
};

constexpr int kMaxRows = 16;

template <bool INT8>
__global__ __launch_bounds__(512) void dyn8_kernel(Dyn8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int stride = p.K + 16;                                       // bytes between rows of codes
  char* xq = smem;                                                   // [M][K + 16]
  float* wmax = reinterpret_cast<float*>(smem + p.M * stride);       // [nwaves][16] per-wave row maxima
  float* rs = wmax + nwaves * kMaxRows;                              // [16] row scales
  float* red = rs + kMaxRows;                                        // [nwaves][256] split-K partials

  // ---- 1. per-row amax -> scale (every workgroup, redundantly: the activation is L2-resident and tiny)
  const int nvec = p.K >> 3;  // 8 bf16 per 16 B
  for (int r = 0; r < p.M; ++r) {
    const u32x4* xr = reinterpret_cast<const u32x4*>(p.x + (size_t)r * p.K);
    float m = 0.f;
    bool has_nan = false;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) m = fmaxf(m, amax8(xr[i], has_nan));
    if (has_nan) m = INFINITY;  // (NaN rows are outside the contract, as in the stand-alone cast)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) wmax[wave * kMaxRows + r] = m;
  }
  __syncthreads();
  if (threadIdx.x < p.M) {
    float m = 0.f;
    for (int w = 0; w < nwaves; ++w) m = fmaxf(m, wmax[w * kMaxRows + threadIdx.x]);
    rs[threadIdx.x] = INT8 ? int8_row_scale(m) : fp8_row_scale(m);
  }
  __syncthreads();
  // ---- 2. cast into LDS
  for (int r = 0; r < p.M; ++r) {
    const u32x4* xr = reinterpret_cast<const u32x4*>(p.x + (size_t)r * p.K);
    const float s = rs[r];
    const float inv = 1.0f / s;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x)
      *reinterpret_cast<u32x2*>(xq + r * stride + i * 8) = INT8 ? int8_quant8(xr[i], inv) : fp8_quant8(xr[i], s);
  }
  __syncthreads();

  // ---- 3. stream this n-tile's weights past the codes
  const int ntile = blockIdx.x;
  const int ksteps = p.K >> 7;  // 128 k per step
  const int ks0 = (ksteps * wave) / nwaves, ks1 = (ksteps * (wave + 1)) / nwaves;
  const int n = ntile * 16 + (lane & 15);
  const int kq = lane >> 4;
  // operand layout of the K = 128 step (stream8_kernels.hip): lane group kq holds k = 16 kq .. +15 and 64 + 16 kq .. +15
  const uint8_t* brow = p.b + (size_t)n * p.K + kq * 16;
  const bool valid = (lane & 15) < p.M;
  const char* arow = xq + (valid ? (lane & 15) : 0) * stride + kq * 16;  // rows past M read row 0 (broadcast) and are zeroed
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // int8: int32 bit patterns
  for (int ks = ks0; ks < ks1; ++ks) {
    const u32x4 b0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(brow + (size_t)ks * 128));
    const u32x4 b1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(brow + (size_t)ks * 128 + 64));
    u32x4 a0 = *reinterpret_cast<const u32x4*>(arow + ks * 128);
    u32x4 a1 = *reinterpret_cast<const u32x4*>(arow + ks * 128 + 64);
    if (!valid) { a0 = u32x4{0, 0, 0, 0}; a1 = u32x4{0, 0, 0, 0}; }
    if constexpr (INT8) {
      i32x4 c = __builtin_bit_cast(i32x4, acc);
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, b1), c, 0, 0, 0);
      acc = __builtin_bit_cast(f32x4, c);
    } else {
      const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
      const i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc, 0, 0, 0, 127, 0, 127);
    }
  }

  // ---- 4. split-K reduction across waves (wave order: reproducible), scales, store
  {
    float* r = red + (size_t)wave * 256 + (kq * 4) * 16 + (lane & 15);  // [row 16][col 16]
    r[0] = acc.x; r[16] = acc.y; r[32] = acc.z; r[48] = acc.w;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 256; idx += blockDim.x) {
    const int row = idx >> 4, col = idx & 15;
    if (row >= p.M) continue;
    const int gn = ntile * 16 + col;
    float v;
    if constexpr (INT8) {
      int isum = 0;
      for (int w = 0; w < nwaves; ++w) isum += __float_as_int(red[(size_t)w * 256 + idx]);
      // t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))   (int8_tensor.py:315-359)
      v = round_bf16((float)isum * rs[row]) * p.col_scale[gn];
    } else {
      float sum = 0.f;
      for (int w = 0; w < nwaves; ++w) sum += red[(size_t)w * 256 + idx];
      v = sum * rs[row] * p.col_scale[gn];
    }
    if (p.bias != nullptr) v += bf16_lo_to_f32(p.bias[gn]);
    p.out[(size_t)row * p.N + gn] = f32_to_bf16_bits(v);
  }
}

template <bool INT8>
int launch_dyn8(const Dyn8Args& p, hipStream_t stream) {
  const int ksteps = p.K >> 7;
  int wpb = 1;
  while (wpb < 8 && ksteps / (wpb * 2) >= 2) wpb *= 2;
  const size_t smem = (size_t)p.M * (p.K + 16) + (size_t)(wpb * kMaxRows + kMaxRows + wpb * 256) * sizeof(float);
  auto kern = dyn8_kernel<INT8>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(dyn8_kernel)")) return rc;
  ao::launch(kern, dim3((unsigned)(p.N / 16)), dim3(wpb * 64), smem, stream, p);
  AO_LAUNCH_CHECK("dyn8_kernel launch");
  return AO_OK;
}

int check_dyn(const char* fn, int64_t M, int64_t N, int64_t K) {
  AO_REQUIRE(M >= 0 && N > 0 && K > 0, "%s: bad shape M=%lld N=%lld K=%lld", fn, (long long)M, (long long)N, (long long)K);
  AO_REQUIRE(N % 16 == 0 && K % 128 == 0, "%s: N=%lld must be a multiple of 16 and K=%lld of 128", fn, (long long)N, (long long)K);
  AO_REQUIRE((M <= kMaxRows && M * (K + 16) <= 64 * 1024) || (M > kMaxRows && mid8_takes_fused(M, N, K)),
             "%s: the fused form holds the cast activation in LDS: M <= 16 and M * (K + 16) <= 65536 (or 16 < M <= 256 on a weight with "
             "few output tiles and K %% 512 == 0), got M=%lld K=%lld (use the cast + matmul entry points)", fn, (long long)M, (long long)K);
  AO_REQUIRE(N < (1ll << 31) && K < (1ll << 31), "%s: dimension too large", fn);
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int ao_dyn_linear_fits(int64_t M, int64_t N, int64_t K) {
  if (M > kMaxRows) return mid8_takes_fused(M, N, K) ? 1 : 0;
  return (M > 0 && M <= kMaxRows && M * (K + 16) <= 64 * 1024 && N % 16 == 0 && K % 128 == 0) ? 1 : 0;
}

extern "C" int ao_int8_dynamic_linear(const uint16_t* x, const int8_t* wq, const float* w_scale, const uint16_t* bias, uint16_t* y,
                                      int64_t M, int64_t N, int64_t K, void* stream) {
  if (int rc = check_dyn(__func__, M, N, K)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(wq);
  AO_REQUIRE_PTR(w_scale);
  AO_REQUIRE_PTR(y);
  if (M > kMaxRows) return mid8_dynamic(true, x, wq, w_scale, bias, y, M, N, K, (hipStream_t)stream);
  if (dec8_takes(M, N, K)) return dec8_dynamic(true, x, wq, w_scale, bias, y, M, N, K, (hipStream_t)stream);
  Dyn8Args p{x, reinterpret_cast<const uint8_t*>(wq), w_scale, bias, y, (int)M, (int)N, (int)K};
  return launch_dyn8<true>(p, (hipStream_t)stream);
}

extern "C" int ao_fp8_dynamic_linear(const uint16_t* x, const uint8_t* wq, const float* w_scale, const uint16_t* bias, uint16_t* y,
                                     int64_t M, int64_t N, int64_t K, void* stream) {
  if (int rc = check_dyn(__func__, M, N, K)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(wq);
  AO_REQUIRE_PTR(w_scale);
  AO_REQUIRE_PTR(y);
  if (M > kMaxRows) return mid8_dynamic(false, x, wq, w_scale, bias, y, M, N, K, (hipStream_t)stream);
  if (dec8_takes(M, N, K)) return dec8_dynamic(false, x, wq, w_scale, bias, y, M, N, K, (hipStream_t)stream);
  Dyn8Args p{x, wq, w_scale, bias, y, (int)M, (int)N, (int)K};
  return launch_dyn8<false>(p, (hipStream_t)stream);
}
