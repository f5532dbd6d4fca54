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
__global__ __launch_bounds__(kThreads) void int8_quant_rowwise_kernel(const uint16_t* __restrict__ x,
                                                                      int8_t* __restrict__ q,
                                                                      float* __restrict__ scale, int64_t K) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * K);
  const int64_t nvec = K >> 3;  // 8 bf16 per 16 B
  float m = 0.f;
  bool has_nan = false;
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) m = fmaxf(m, amax8(xr[i], has_nan));
  m = block_max(has_nan ? INFINITY : m, red);  // (NaN rows are outside the contract)
  const float s = int8_row_scale(m);
  const float inv = 1.0f / s;
  if (threadIdx.x == 0) scale[row] = s;
  u32x2* qr = reinterpret_cast<u32x2*>(q + row * K);
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) qr[i] = int8_quant8(xr[i], inv);
}

// ---- fp8 e4m3fn per-row ----------------------------------------------------------
// scale = f32(bf16(amax / 448));  q = e4m3_rne(clamp(f32(x) / scale, -448, 448))
__global__ __launch_bounds__(kThreads) void fp8_quant_rowwise_kernel(const uint16_t* __restrict__ x,
                                                                     uint8_t* __restrict__ q,
                                                                     float* __restrict__ scale, int64_t K) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * K);
  const int64_t nvec = K >> 3;
  float m = 0.f;
  bool has_nan = false;
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) m = fmaxf(m, amax8(xr[i], has_nan));
  m = block_max(has_nan ? INFINITY : m, red);
  const float s = fp8_row_scale(m);
  if (threadIdx.x == 0) scale[row] = s;
  u32x2* qr = reinterpret_cast<u32x2*>(q + row * K);
  for (int64_t i = threadIdx.x; i < nvec; i += kThreads) qr[i] = fp8_quant8(xr[i], s);
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
  bool has_nan = false;
  float m = amax8(v, has_nan);
  m = fmaxf(m, __shfl_xor(m, 1));
  m = fmaxf(m, __shfl_xor(m, 2));
  uint32_t nanbits = has_nan ? 1u : 0u;
  nanbits |= __shfl_xor(nanbits, 1);
  nanbits |= __shfl_xor(nanbits, 2);
  const bool finite = (nanbits == 0u) && (m < INFINITY);
  uint32_t e;  // biased E8M0 exponent
  if (MODE == AO_MX_SCALE_RCEIL) {
    // descale = amax * (1/448) in fp32; round its value up to a power of two
    const uint32_t bits = f32_to_bits(m * (1.0f / 448.0f));
    const uint32_t be = (bits >> 23) & 0xffu, mant = bits & 0x7fffffu;
    const bool up = (be == 0) ? (mant > 0x400000u) : (mant != 0);
    e = be + (up ? 1u : 0u);
  } else {
    // floor(log2(amax)) - 8, clamped to [-127, 128], biased
    const int ex = (int)((f32_to_bits(m) >> 23) & 0xffu) - 127 - 8;
    e = (uint32_t)(min(max(ex, -127), 128) + 127);
  }
  if (!finite) e = 255u;
  // reciprocal scale 2^(127 - e) built from the E8M0 byte 254 - e (mx_tensor.py:132-158)
  const uint32_t re = (254u - e) & 0xffu;
  uint32_t rbits = re << 23;
  if (re == 0u) rbits = 0x00400000u;    // 2^-127 as an fp32 subnormal
  if (re == 255u) rbits = 0x7F800001u;  // NaN
  const float r = bits_to_f32(rbits);
  float f[8] = {bf16_lo_to_f32(v.x), bf16_hi_to_f32(v.x), bf16_lo_to_f32(v.y), bf16_hi_to_f32(v.y),
                bf16_lo_to_f32(v.z), bf16_hi_to_f32(v.z), bf16_lo_to_f32(v.w), bf16_hi_to_f32(v.w)};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f[j] *= r;
    if (MODE == AO_MX_SCALE_FLOOR) f[j] = clamp448(f[j]);  // eager saturation (torch < 2.13), :361-373
  }
  reinterpret_cast<u32x2*>(q + blk * 32)[part] =
      u32x2{cvt4_e4m3(f[0], f[1], f[2], f[3]), cvt4_e4m3(f[4], f[5], f[6], f[7])};
  if (part == 0) scale[blk] = (uint8_t)e;
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
  ao::launch(int8_quant_rowwise_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, q, scale, K);
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
  ao::launch(fp8_quant_rowwise_kernel, dim3((unsigned)M), dim3(kThreads), 0, (hipStream_t)stream, x, q, scale, K);
  AO_LAUNCH_CHECK("fp8_quant_rowwise_kernel launch");
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
