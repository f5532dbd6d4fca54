// Activation / weight quantisation ("fused prologue") kernels for gfx950:
// per-row int8, per-row float8 e4m3fn (OCP), and MXFP8 (1x32 blocks, E8M0 scales).
// Each replays the reference's op sequence exactly (same intermediate dtypes and
// roundings) in ONE pass over HBM: the row is read once (kept L2-hot for the
// second sweep), reduced, and written once as 1 B/element.
//
// Reference (torchao 0.19.0 snapshot):
//   int8 : quantization/quantize_/workflows/int8/int8_tensor.py:191-230,
//          quant_primitives.py:1534-1583 (choose_qparams_affine), :463-485
//   fp8  : quantize_/workflows/float8/float8_tensor.py:167-253,
//          quant_primitives.py:2192-2212, 2271-2287
//   mx   : prototype/mx_formats/mx_tensor.py:111-225 (RCEIL), :228-409 (to_mx)
#include <algorithm>

#include "common.h"
#include "quant_math.h"

namespace ao {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float block_max(float v, float* red) {
  // wave reduce then across the 4 waves of the block
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}

// ---- int8 per-row symmetric ----------------------------------------------------
// scale = f32(max(bf16(amax / 127.5), bf16(f32_eps)));  q = clamp(rint(x * (1/scale)), -128, 127)
// `amax_in` (optional): the row's amax over the FULL K when x is only a K shard of the activation (row-parallel TP linears:
// the scale must be the unsharded one); `ldx`: row stride of x in elements.
__global__ __launch_bounds__(kThreads) void int8_quant_rowwise_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                      const float* __restrict__ amax_in,
                                                                      int8_t* __restrict__ q,
                                                                      float* __restrict__ scale, int64_t K) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * ldx);
  const int64_t nvec = K >> 3;  // 8 bf16 per 16 B
  float m = 0.f;
  if (amax_in != nullptr) {
    m = amax_in[row];
  } else {
    bool has_nan = false;
    for (int64_t i = threadIdx.x; i < nvec; i += kThreads) m = fmaxf(m, amax8(xr[i], has_nan));
    m = block_max(has_nan ? INFINITY : m, red);  // (NaN rows are outside the contract)
  }
  const float s = int8_row_scale(m);
  const float inv = 1.0f / s;
  if (threadIdx.x == 0) scale[row] = s;
  u32x2* qr = reinterpret_cast<u32x2*>(q + row * K);
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) qr[i] = int8_quant8(xr[i], inv);
}

// ---- fp8 e4m3fn per-row ----------------------------------------------------------
// scale = f32(bf16(amax / 448));  q = e4m3_rne(clamp(f32(x) / scale, -448, 448))
__global__ __launch_bounds__(kThreads) void fp8_quant_rowwise_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ amax_in,
                                                                     uint8_t* __restrict__ q,
                                                                     float* __restrict__ scale, int64_t K) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * ldx);
  const int64_t nvec = K >> 3;
  float m = 0.f;
  if (amax_in != nullptr) {
    m = amax_in[row];
  } else {
    bool has_nan = false;
    for (int64_t i = threadIdx.x; i < nvec; i += kThreads) m = fmaxf(m, amax8(xr[i], has_nan));
    m = block_max(has_nan ? INFINITY : m, red);
  }
  const float s = fp8_row_scale(m);
  if (threadIdx.x == 0) scale[row] = s;
  u32x2* qr = reinterpret_cast<u32x2*>(q + row * K);
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) qr[i] = fp8_quant8(xr[i], s);
}

// ---- the same two casts with the row held in registers between the amax reduction and the cast (K <= 16384): one read of x
// instead of a second sweep through L2.  NV = 16-byte vectors per thread.
template <bool INT8, int NV>
__global__ __launch_bounds__(kThreads) void quant_rowwise_reg_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ amax_in, uint8_t* __restrict__ q,
                                                                     float* __restrict__ scale, int64_t K) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * ldx);
  const int nvec = (int)(K >> 3);
  u32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    v[i] = (idx < nvec) ? xr[idx] : u32x4{0u, 0u, 0u, 0u};
  }
  float m = 0.f;
  if (amax_in != nullptr) {
    m = amax_in[row];
  } else {
    bool has_nan = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) m = fmaxf(m, amax8(v[i], has_nan));
    m = block_max(has_nan ? INFINITY : m, red);  // (NaN rows are outside the contract)
  }
  const float s = INT8 ? int8_row_scale(m) : fp8_row_scale(m);
  const float inv = 1.0f / s;
  if (threadIdx.x == 0) scale[row] = s;
  u32x2* qr = reinterpret_cast<u32x2*>(q + row * K);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    if (idx < nvec) qr[idx] = INT8 ? int8_quant8(v[i], inv) : fp8_quant8(v[i], s);
  }
}

template <bool INT8>
bool launch_quant_rowwise_reg(const uint16_t* x, int64_t ldx, const float* amax, void* q, float* scale, int64_t M, int64_t K, hipStream_t s) {
  const int64_t per_thread = ((K >> 3) + kThreads - 1) / kThreads;
  const dim3 grid((unsigned)M), block(kThreads);
  uint8_t* qb = reinterpret_cast<uint8_t*>(q);
  if (per_thread <= 1) ao::launch(quant_rowwise_reg_kernel<INT8, 1>, grid, block, 0, s, x, ldx, amax, qb, scale, K);
  else if (per_thread <= 2) ao::launch(quant_rowwise_reg_kernel<INT8, 2>, grid, block, 0, s, x, ldx, amax, qb, scale, K);
  else if (per_thread <= 4) ao::launch(quant_rowwise_reg_kernel<INT8, 4>, grid, block, 0, s, x, ldx, amax, qb, scale, K);
  else if (per_thread <= 8) ao::launch(quant_rowwise_reg_kernel<INT8, 8>, grid, block, 0, s, x, ldx, amax, qb, scale, K);
  else return false;  // longer rows: the two-sweep kernels
  return true;
}

// ---- per-row amax of a (possibly strided) bf16 matrix: the local half of a full-K activation scale under K sharding -----
__global__ __launch_bounds__(kThreads) void rowwise_amax_kernel(const uint16_t* __restrict__ x, int64_t ldx, float* __restrict__ amax,
                                                                int64_t K) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * ldx);
  float m = 0.f;
  bool has_nan = false;
  for (int64_t i = threadIdx.x; i < (K >> 3); i += kThreads) m = fmaxf(m, amax8(xr[i], has_nan));
  m = block_max(has_nan ? INFINITY : m, red);
  if (threadIdx.x == 0) amax[row] = m;
}

// ---- scale epilogues over all-reduced accumulators (row-parallel TP linears): the arithmetic of the fused GEMM epilogues ----
// int8: t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))        (int8_tensor.py:315-359)
// fp8 : y = bf16(c * sa[m] * sb[n] (+ bias))                                (float8/inference.py:104-123)
template <bool INT8>
__global__ __launch_bounds__(kThreads) void scale_epilogue_kernel(const void* __restrict__ acc, const float* __restrict__ row_scale,
                                                                  const float* __restrict__ col_scale, const uint16_t* __restrict__ bias,
                                                                  uint16_t* __restrict__ y, int64_t N) {
  const int64_t row = blockIdx.y;
  const float rs = row_scale[row];
  for (int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x; n < N; n += (int64_t)gridDim.x * kThreads) {
    float v;
    if (INT8) {
      const float t = round_bf16((float)reinterpret_cast<const int32_t*>(acc)[row * N + n] * rs);
      v = t * col_scale[n];
    } else {
      v = reinterpret_cast<const float*>(acc)[row * N + n] * rs * col_scale[n];
    }
    if (bias != nullptr) v += bf16_lo_to_f32(bias[n]);
    y[row * N + n] = f32_to_bf16_bits(v);
  }
}

// ---- int8 per-row ASYMMETRIC (Int8DynamicActivationInt8WeightConfig(act_mapping_type=ASYMMETRIC)) --------------------
// choose_qparams_affine's ASYMMETRIC branch on a bf16 row (quant_primitives.py:1568-1574; every tensor op rounds to bf16):
//   mn = min(min(row), 0), mx = max(max(row), 0)
//   scale = max(bf16(bf16(mx - mn) / 255), bf16(f32_eps));  zp = clamp(-128 - rint(bf16(mn / scale)), -128, 127)
//   q = clamp(rint(x * (1 / scale)) + zp, -128, 127)                                       (quantize_affine, :463-485)
__global__ __launch_bounds__(kThreads) void int8_quant_rowwise_asym_kernel(const uint16_t* __restrict__ x, int8_t* __restrict__ q,
                                                                           float* __restrict__ scale, int8_t* __restrict__ zero_point,
                                                                           int64_t K) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * K);
  const int64_t nvec = K >> 3;
  float mx = 0.f, mn = 0.f;  // the zero each is clamped against is the identity here
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) {
    const u32x4 v = xr[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16_lo_to_f32(w[j]), b = bf16_hi_to_f32(w[j]);
      mx = fmaxf(mx, fmaxf(a, b));
      mn = fminf(mn, fminf(a, b));
    }
  }
  mx = block_max(mx, red);
  mn = -block_max(-mn, red);
  const float s = fmaxf(round_bf16(round_bf16(mx - mn) / 255.0f), 1.1920928955078125e-07f);
  const float zp = fminf(fmaxf(-128.0f - rintf(round_bf16(mn / s)), -128.0f), 127.0f);
  const float inv = 1.0f / s;
  if (threadIdx.x == 0) {
    scale[row] = s;
    zero_point[row] = (int8_t)(int)zp;
  }
  u32x2* qr = reinterpret_cast<u32x2*>(q + row * K);
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) {
    const u32x4 v = xr[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t out[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = fminf(fmaxf(rintf(bf16_lo_to_f32(w[j]) * inv) + zp, -128.f), 127.f);
      const float b = fminf(fmaxf(rintf(bf16_hi_to_f32(w[j]) * inv) + zp, -128.f), 127.f);
      out[j >> 1] |= (((uint32_t)(int)a & 0xffu) | (((uint32_t)(int)b & 0xffu) << 8)) << ((j & 1) * 16);
    }
    qr[i] = u32x2{out[0], out[1]};
  }
}

// static activation quantization (Int8StaticActivationInt8WeightConfig: Int8Tensor.from_hp(x, scale=..., zero_point=...), int8_tensor.py:
// 212-231): q = clamp(rint(x * (1 / scale)) + zp, -128, 127) with the GIVEN scale / zero-point, one per tensor (stride 0) or per row
__global__ __launch_bounds__(kThreads) void int8_quant_static_kernel(const uint16_t* __restrict__ x, const float* __restrict__ scale,
                                                                     const int8_t* __restrict__ zero_point, int stride,
                                                                     int8_t* __restrict__ q, int64_t K) {
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * K);
  const float inv = 1.0f / scale[row * stride];
  const float zp = zero_point != nullptr ? (float)zero_point[row * stride] : 0.0f;
  u32x2* qr = reinterpret_cast<u32x2*>(q + row * K);
  for (int64_t i = threadIdx.x; i < (K >> 3); i += kThreads) {
    const u32x4 v = xr[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t out[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = fminf(fmaxf(rintf(bf16_lo_to_f32(w[j]) * inv) + zp, -128.f), 127.f);
      const float b = fminf(fmaxf(rintf(bf16_hi_to_f32(w[j]) * inv) + zp, -128.f), 127.f);
      out[j >> 1] |= (((uint32_t)(int)a & 0xffu) | (((uint32_t)(int)b & 0xffu) << 8)) << ((j & 1) * 16);
    }
    qr[i] = u32x2{out[0], out[1]};
  }
}

// row sums of an int8 [N][K] weight (the zero-point correction's rowsum(W), int8_tensor.py:326): one wave per row
__global__ __launch_bounds__(kThreads) void int8_row_sums_kernel(const int8_t* __restrict__ q, int32_t* __restrict__ sums, int64_t N, int64_t K) {
  const int64_t row = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  const u32x4* qr = reinterpret_cast<const u32x4*>(q + row * K);
  int32_t acc = 0;
  for (int64_t i = lane; i < (K >> 4); i += 64) {
    const u32x4 v = qr[i];
    acc = __builtin_amdgcn_sdot4((int)v.x, 0x01010101, acc, false);
    acc = __builtin_amdgcn_sdot4((int)v.y, 0x01010101, acc, false);
    acc = __builtin_amdgcn_sdot4((int)v.z, 0x01010101, acc, false);
    acc = __builtin_amdgcn_sdot4((int)v.w, 0x01010101, acc, false);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) sums[row] = acc;
}

// the Int8Tensor linear's epilogue with an asymmetric activation (int8_tensor.py:305-346):
//   t = bf16(f32(c) * sx[m]);  corr = bf16((f32(zp[m]) * sx[m]) * f32(wsum[n]));  t = bf16(t - corr);  y = bf16(t * sw[n] (+ bias))
__global__ __launch_bounds__(kThreads) void scale_epilogue_asym_kernel(const int32_t* __restrict__ acc, const float* __restrict__ x_scale,
                                                                       const int8_t* __restrict__ x_zp, const int32_t* __restrict__ w_sums,
                                                                       const float* __restrict__ w_scale, const uint16_t* __restrict__ bias,
                                                                       uint16_t* __restrict__ y, int64_t N) {
  const int64_t row = blockIdx.y;
  const float xs = x_scale[row];
  const float zs = (float)x_zp[row] * xs;
  for (int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x; n < N; n += (int64_t)gridDim.x * kThreads) {
    float t = round_bf16((float)acc[row * N + n] * xs);
    t = round_bf16(t - round_bf16(zs * (float)w_sums[n]));
    float v = t * w_scale[n];
    if (bias != nullptr) v += bf16_lo_to_f32(bias[n]);
    y[row * N + n] = f32_to_bf16_bits(v);
  }
}

// ---- MXFP8: one E8M0 scale per 32 elements along the row -------------------------
// 4 lanes own one 32-element block (8 elements each); 64 blocks per 256-thread group.
template <int MODE>  // AO_MX_SCALE_FLOOR / AO_MX_SCALE_RCEIL
__global__ __launch_bounds__(kThreads) void mxfp8_quant_kernel(const uint16_t* __restrict__ x,
                                                               uint8_t* __restrict__ q,
                                                               uint8_t* __restrict__ scale,
                                                               int64_t total_blocks) {
  const int64_t blk = (int64_t)blockIdx.x * (kThreads / 4) + (threadIdx.x >> 2);
  if (blk >= total_blocks) return;  // whole 4-lane groups exit together
  const int part = threadIdx.x & 3;
  const u32x4 v = reinterpret_cast<const u32x4*>(x + blk * 32)[part];
  uint32_t e;  // biased E8M0 exponent
  reinterpret_cast<u32x2*>(q + blk * 32)[part] = mx_cast8<MODE>(v, e);  // quant_math.h: shared with the grouped GEMM's fused cast
  if (part == 0) scale[blk] = (uint8_t)e;
}

// ---- MXFP8 colwise: one E8M0 scale per 32 elements along the ROWS (32 x 1 blocks), data written column-major -------------
// (mxfp8_quantize.cuh:460-820 colwise branch; torch reference to_mx(x.t()).t()).  A 4-wave workgroup owns a 128 x 128 tile:
// wave w owns the row block 32 w .. 32 w + 31 and a lane two adjacent columns -- the 32 rows of a column are 32 registers of
// ONE lane, so amax and scale need no cross-lane step; rows are read as coalesced 256-byte pieces.  The codes go through LDS
// (column stride 132 B: 2-way conflicts) so that every column leaves as one 128-byte run of the transposed output.
constexpr int kColTile = 128;
constexpr int kColLdsStride = 132;
template <int MODE>
__global__ __launch_bounds__(kThreads) void mxfp8_quant_colwise_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ qt,
                                                                       uint8_t* __restrict__ scale, int64_t R, int64_t C) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[kColTile * kColLdsStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // blockIdx.z: the matrix of a batch (expert) -- each [R][C] matrix is cast on its own, outputs laid out per matrix
  x += (int64_t)blockIdx.z * R * C;
  qt += (int64_t)blockIdx.z * R * C;
  scale += (int64_t)blockIdx.z * (R >> 5) * C;
  const int64_t r0 = (int64_t)blockIdx.y * kColTile, c0 = (int64_t)blockIdx.x * kColTile;
  const int64_t rb = r0 + wave * 32, c = c0 + 2 * lane;
  if (rb < R && c < C) {
    uint32_t v[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = *reinterpret_cast<const uint32_t*>(x + (rb + r) * C + c);
    float m0 = 0.f, m1 = 0.f;
    bool nan0 = false, nan1 = false;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const float a = fabsf(bf16_lo_to_f32(v[r])), b = fabsf(bf16_hi_to_f32(v[r]));
      nan0 |= (a != a); nan1 |= (b != b);
      m0 = fmaxf(m0, a); m1 = fmaxf(m1, b);
    }
    const uint32_t e0 = mx_block_exponent<MODE>(m0, !nan0 && m0 < INFINITY), e1 = mx_block_exponent<MODE>(m1, !nan1 && m1 < INFINITY);
    const float q0 = mx_reciprocal(e0), q1 = mx_reciprocal(e1);
    uint32_t* t0 = reinterpret_cast<uint32_t*>(tile + (2 * lane) * kColLdsStride + wave * 32);
    uint32_t* t1 = reinterpret_cast<uint32_t*>(tile + (2 * lane + 1) * kColLdsStride + wave * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = bf16_lo_to_f32(v[4 * i + j]) * q0;
        b[j] = bf16_hi_to_f32(v[4 * i + j]) * q1;
        if (MODE == AO_MX_SCALE_FLOOR) { a[j] = clamp448(a[j]); b[j] = clamp448(b[j]); }
      }
      t0[i] = cvt4_e4m3(a[0], a[1], a[2], a[3]);
      t1[i] = cvt4_e4m3(b[0], b[1], b[2], b[3]);
    }
    *reinterpret_cast<uint16_t*>(scale + (rb >> 5) * C + c) = (uint16_t)(e0 | (e1 << 8));
  }
  __syncthreads();
  // 128 columns x 8 pieces of 16 bytes (= 16 rows each)
  for (int p = threadIdx.x; p < kColTile * 8; p += kThreads) {
    const int col = p >> 3, part = p & 7;
    const int64_t gc = c0 + col, gr = r0 + part * 16;
    if (gc < C && gr < R) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(tile + col * kColLdsStride + part * 16);
      *reinterpret_cast<u32x4*>(qt + gc * R + gr) = u32x4{src[0], src[1], src[2], src[3]};
    }
  }
}

int check_rows(const char* fn, int64_t M, int64_t K, int64_t mult) {
  AO_REQUIRE(M >= 0 && K > 0, "%s: bad shape M=%lld K=%lld", fn, (long long)M, (long long)K);
  AO_REQUIRE(K % mult == 0, "%s: K=%lld must be a multiple of %lld", fn, (long long)K, (long long)mult);
  AO_REQUIRE(M < (1ll << 31), "%s: M=%lld too large", fn, (long long)M);
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int ao_int8_quantize_rowwise(const uint16_t* x, int8_t* q, float* scale, int64_t M, int64_t K,
                                        void* stream) {
  if (int rc = check_rows(__func__, M, K, 8)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(q);
  AO_REQUIRE_PTR(scale);
  if (!launch_quant_rowwise_reg<true>(x, K, nullptr, q, scale, M, K, (hipStream_t)stream))
    ao::launch(int8_quant_rowwise_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, K, (const float*)nullptr, q, scale, K);
  AO_LAUNCH_CHECK("int8_quant_rowwise_kernel launch");
  return AO_OK;
}

extern "C" int ao_fp8_quantize_rowwise(const uint16_t* x, uint8_t* q, float* scale, int64_t M, int64_t K,
                                       void* stream) {
  if (int rc = check_rows(__func__, M, K, 8)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(q);
  AO_REQUIRE_PTR(scale);
  if (!launch_quant_rowwise_reg<false>(x, K, nullptr, q, scale, M, K, (hipStream_t)stream))
    ao::launch(fp8_quant_rowwise_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, K, (const float*)nullptr, q, scale, K);
  AO_LAUNCH_CHECK("fp8_quant_rowwise_kernel launch");
  return AO_OK;
}

namespace {
int mxfp8_colwise(const char* fn, const uint16_t* x, uint8_t* q_t, uint8_t* scale_e8m0, int64_t E, int64_t R, int64_t C, int scaling_mode,
                  void* stream) {
  if (int rc = check_rows(fn, R, C, 32)) return rc;
  AO_REQUIRE(R % 32 == 0, "%s: R=%lld must be a multiple of 32", fn, (long long)R);
  AO_REQUIRE(E >= 0 && E <= 65535, "%s: E=%lld must be in [0, 65535]", fn, (long long)E);
  AO_REQUIRE(scaling_mode == AO_MX_SCALE_FLOOR || scaling_mode == AO_MX_SCALE_RCEIL,
             "%s: scaling_mode must be AO_MX_SCALE_FLOOR or AO_MX_SCALE_RCEIL, got %d", fn, scaling_mode);
  if (R == 0 || E == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(q_t);
  AO_REQUIRE_PTR(scale_e8m0);
  const int64_t gx = (C + kColTile - 1) / kColTile, gy = (R + kColTile - 1) / kColTile;
  AO_REQUIRE(gy <= 65535, "%s: R=%lld too large for one launch", fn, (long long)R);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)E);
  if (scaling_mode == AO_MX_SCALE_RCEIL)
    ao::launch(mxfp8_quant_colwise_kernel<AO_MX_SCALE_RCEIL>, grid, dim3(kThreads), 0, s, x, q_t, scale_e8m0, R, C);
  else
    ao::launch(mxfp8_quant_colwise_kernel<AO_MX_SCALE_FLOOR>, grid, dim3(kThreads), 0, s, x, q_t, scale_e8m0, R, C);
  AO_LAUNCH_CHECK("mxfp8_quant_colwise_kernel launch");
  return AO_OK;
}
}  // namespace

extern "C" int ao_mxfp8_quantize_colwise(const uint16_t* x, uint8_t* q_t, uint8_t* scale_e8m0, int64_t R, int64_t C, int scaling_mode,
                                         void* stream) {
  return mxfp8_colwise(__func__, x, q_t, scale_e8m0, 1, R, C, scaling_mode, stream);
}

extern "C" int ao_mxfp8_quantize_colwise_3d(const uint16_t* x, uint8_t* q_t, uint8_t* scale_e8m0, int64_t E, int64_t R, int64_t C,
                                            int scaling_mode, void* stream) {
  return mxfp8_colwise(__func__, x, q_t, scale_e8m0, E, R, C, scaling_mode, stream);
}

namespace {
int check_strided(const char* fn, int64_t M, int64_t K, int64_t ldx) {
  if (int rc = check_rows(fn, M, K, 8)) return rc;
  AO_REQUIRE(ldx >= K && ldx % 8 == 0, "%s: row stride %lld must be >= K=%lld and a multiple of 8 elements", fn, (long long)ldx, (long long)K);
  return AO_OK;
}
}  // namespace

extern "C" int ao_rowwise_amax(const uint16_t* x, int64_t ldx, float* amax, int64_t M, int64_t K, void* stream) {
  if (int rc = check_strided(__func__, M, K, ldx)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(amax);
  ao::launch(rowwise_amax_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, amax, K);
  AO_LAUNCH_CHECK("rowwise_amax_kernel launch");
  return AO_OK;
}

extern "C" int ao_int8_quantize_rowwise_amax(const uint16_t* x, int64_t ldx, const float* amax, int8_t* q, float* scale, int64_t M,
                                             int64_t K, void* stream) {
  if (int rc = check_strided(__func__, M, K, ldx)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(amax);
  AO_REQUIRE_PTR(q);
  AO_REQUIRE_PTR(scale);
  if (!launch_quant_rowwise_reg<true>(x, ldx, amax, q, scale, M, K, (hipStream_t)stream))
    ao::launch(int8_quant_rowwise_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, amax, q, scale, K);
  AO_LAUNCH_CHECK("int8_quant_rowwise_kernel launch");
  return AO_OK;
}

extern "C" int ao_fp8_quantize_rowwise_amax(const uint16_t* x, int64_t ldx, const float* amax, uint8_t* q, float* scale, int64_t M,
                                            int64_t K, void* stream) {
  if (int rc = check_strided(__func__, M, K, ldx)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(amax);
  AO_REQUIRE_PTR(q);
  AO_REQUIRE_PTR(scale);
  if (!launch_quant_rowwise_reg<false>(x, ldx, amax, q, scale, M, K, (hipStream_t)stream))
    ao::launch(fp8_quant_rowwise_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, amax, q, scale, K);
  AO_LAUNCH_CHECK("fp8_quant_rowwise_kernel launch");
  return AO_OK;
}

namespace {
template <bool INT8>
int scale_epilogue(const char* fn, const void* acc, const float* row_scale, const float* col_scale, const uint16_t* bias, uint16_t* y,
                   int64_t M, int64_t N, void* stream) {
  AO_REQUIRE(M >= 0 && N > 0 && M <= 65535ll * 65535ll, "%s: bad shape M=%lld N=%lld", fn, (long long)M, (long long)N);
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(acc);
  AO_REQUIRE_PTR(row_scale);
  AO_REQUIRE_PTR(col_scale);
  AO_REQUIRE_PTR(y);
  for (int64_t m0 = 0; m0 < M; m0 += 65535) {
    const int64_t rows = std::min<int64_t>(65535, M - m0);
    const char* a = reinterpret_cast<const char*>(acc) + m0 * N * 4;
    dim3 grid((unsigned)std::min<int64_t>((N + kThreads - 1) / kThreads, 64), (unsigned)rows);
    ao::launch(scale_epilogue_kernel<INT8>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const void*)a, row_scale + m0, col_scale, bias,
               y + m0 * N, N);
  }
  AO_LAUNCH_CHECK("scale_epilogue_kernel launch");
  return AO_OK;
}
}  // namespace

extern "C" int ao_int8_scale_epilogue(const int32_t* acc, const float* x_scale, const float* w_scale, const uint16_t* bias, uint16_t* y,
                                      int64_t M, int64_t N, void* stream) {
  return scale_epilogue<true>(__func__, acc, x_scale, w_scale, bias, y, M, N, stream);
}

extern "C" int ao_fp8_scale_epilogue(const float* acc, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                                     int64_t M, int64_t N, void* stream) {
  return scale_epilogue<false>(__func__, acc, scale_a, scale_b, bias, y, M, N, stream);
}

extern "C" int ao_int8_quantize_rowwise_asym(const uint16_t* x, int8_t* q, float* scale, int8_t* zero_point, int64_t M, int64_t K,
                                             void* stream) {
  if (int rc = check_rows(__func__, M, K, 8)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(q);
  AO_REQUIRE_PTR(scale);
  AO_REQUIRE_PTR(zero_point);
  ao::launch(int8_quant_rowwise_asym_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, q, scale, zero_point, K);
  AO_LAUNCH_CHECK("int8_quant_rowwise_asym_kernel launch");
  return AO_OK;
}

extern "C" int ao_int8_quantize_static(const uint16_t* x, const float* scale, const int8_t* zero_point, int per_row, int8_t* q, int64_t M,
                                       int64_t K, void* stream) {
  if (int rc = check_rows(__func__, M, K, 8)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(scale);
  AO_REQUIRE_PTR(q);
  ao::launch(int8_quant_static_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, scale, zero_point, per_row ? 1 : 0, q, K);
  AO_LAUNCH_CHECK("int8_quant_static_kernel launch");
  return AO_OK;
}

extern "C" int ao_int8_row_sums(const int8_t* q, int32_t* sums, int64_t N, int64_t K, void* stream) {
  if (int rc = check_rows(__func__, N, K, 16)) return rc;
  if (N == 0) return AO_OK;
  AO_REQUIRE_PTR(q);
  AO_REQUIRE_PTR(sums);
  AO_REQUIRE(K <= (1ll << 24), "%s: K=%lld too large for an int32 row sum", __func__, (long long)K);
  ao::launch(int8_row_sums_kernel, dim3((unsigned)((N + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, q, sums, N, K);
  AO_LAUNCH_CHECK("int8_row_sums_kernel launch");
  return AO_OK;
}

extern "C" int ao_int8_scale_epilogue_asym(const int32_t* acc, const float* x_scale, const int8_t* x_zero_point, const int32_t* w_row_sums,
                                           const float* w_scale, const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, void* stream) {
  AO_REQUIRE(M >= 0 && N > 0 && M <= 65535ll * 65535ll, "%s: bad shape M=%lld N=%lld", __func__, (long long)M, (long long)N);
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(acc);
  AO_REQUIRE_PTR(x_scale);
  AO_REQUIRE_PTR(x_zero_point);
  AO_REQUIRE_PTR(w_row_sums);
  AO_REQUIRE_PTR(w_scale);
  AO_REQUIRE_PTR(y);
  for (int64_t m0 = 0; m0 < M; m0 += 65535) {
    const int64_t rows = std::min<int64_t>(65535, M - m0);
    dim3 grid((unsigned)std::min<int64_t>((N + kThreads - 1) / kThreads, 64), (unsigned)rows);
    ao::launch(scale_epilogue_asym_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, acc + m0 * N, x_scale + m0, x_zero_point + m0,
               w_row_sums, w_scale, bias, y + m0 * N, N);
  }
  AO_LAUNCH_CHECK("scale_epilogue_asym_kernel launch");
  return AO_OK;
}

extern "C" int ao_mxfp8_quantize_rowwise(const uint16_t* x, uint8_t* q, uint8_t* scale_e8m0, int64_t R,
                                         int64_t C, int scaling_mode, void* stream) {
  if (int rc = check_rows(__func__, R, C, 32)) return rc;
  AO_REQUIRE(scaling_mode == AO_MX_SCALE_FLOOR || scaling_mode == AO_MX_SCALE_RCEIL,
             "ao_mxfp8_quantize_rowwise: scaling_mode must be AO_MX_SCALE_FLOOR or AO_MX_SCALE_RCEIL, got %d",
             scaling_mode);
  if (R == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(q);
  AO_REQUIRE_PTR(scale_e8m0);
  const int64_t blocks = R * (C / 32);
  const int64_t grid = (blocks + (kThreads / 4) - 1) / (kThreads / 4);
  AO_REQUIRE(grid < (1ll << 31), "ao_mxfp8_quantize_rowwise: tensor too large for one launch");
  hipStream_t s = (hipStream_t)stream;
  if (scaling_mode == AO_MX_SCALE_RCEIL)
    ao::launch(mxfp8_quant_kernel<AO_MX_SCALE_RCEIL>, dim3((unsigned)grid), dim3(kThreads), 0, s, x, q, scale_e8m0, blocks);
  else
    ao::launch(mxfp8_quant_kernel<AO_MX_SCALE_FLOOR>, dim3((unsigned)grid), dim3(kThreads), 0, s, x, q, scale_e8m0, blocks);
  AO_LAUNCH_CHECK("mxfp8_quant_kernel launch");
  return AO_OK;
}
