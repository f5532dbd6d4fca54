// HQQ (half-quadratic quantization) qparams for the int4 tinygemm format on gfx950: what
// Int4TilePackedTo4dTensor.from_hp(..., int4_choose_qparams_algorithm=HQQ) computes in the reference
// (quantize_/workflows/int4/int4_tile_packed_to_4d_tensor.py:149-168 -> _choose_qparams_and_quantize_affine_hqq,
// quant_primitives.py:1891-1997 -> optimize_weights_proximal_legacy / _shrink_lp_op, :1797-1866 ->
// _convert_to_affinequantized_format, :1874-1888), the README's recommended int4 recipe.
//
// The reference's GPU path optimises in float16 (`dtype = torch.float16 if device.type == "cuda"`): every elementwise op is
// fp32 compute + round to fp16, python scalars stay fp32, the per-group mean and the global error accumulate in fp32.  The
// kernels below replay that op sequence element for element (h() = round to fp16); `__fmul_rn` / `__fadd_rn` keep the
// compiler from contracting the fp32 steps the reference rounds separately.  One wave per group (two groups per wave at g = 32).
// The loop is host-driven like the reference's (`float(err.mean())` synchronises every iteration there too): one launch per
// iteration, early stop on the fp16 global error.
#include "common.h"

namespace ao {
namespace {

__device__ __forceinline__ float h(float x) { return (float)(_Float16)x; }  // fp32 -> fp16 (RNE) -> fp32

template <int G>
struct HqqGeom {
  static constexpr int LANES = (G < 64) ? G : 64;  // lanes per group
  static constexpr int E = G / LANES;              // elements per lane
  static constexpr int GROUPS_PER_WAVE = 64 / LANES;
};

// params: [0 .. groups) scale (fp16 value as float), [groups .. 2 groups) zero
template <int G>
__global__ __launch_bounds__(64) void hqq_init_kernel(const uint16_t* __restrict__ w, float* __restrict__ params, int64_t groups) {
  using Geo = HqqGeom<G>;
  const int lane = threadIdx.x;
  const int64_t grp = (int64_t)blockIdx.x * Geo::GROUPS_PER_WAVE + lane / Geo::LANES;
  const int li = lane % Geo::LANES;
  if (grp >= groups) return;
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < Geo::E; ++j) {
    const float v = bf16_lo_to_f32(w[grp * G + li + j * Geo::LANES]);
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int off = 1; off < Geo::LANES; off <<= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
  if (li == 0) {
    // max_v / (max - min) with a python scalar numerator is reciprocal() * max_v in torch; clamp(max=2e4); zero = round(-min * scale)
    const float scale = fminf(__fmul_rn(__frcp_rn(__fsub_rn(mx, mn)), 15.0f), 2e4f);
    const float zero = rintf(__fmul_rn(-mn, scale));
    params[grp] = h(scale);
    params[groups + grp] = h(zero);
  }
}

// one proximal iteration: zero <- mean(W_q - (W_f - W_e) * scale); err += sum |W_f - W_r|
template <int G>
__global__ __launch_bounds__(64) void hqq_iter_kernel(const uint16_t* __restrict__ w, float* __restrict__ params, double* __restrict__ err,
                                                      int64_t groups, float inv_beta) {
  using Geo = HqqGeom<G>;
  const int lane = threadIdx.x;
  const int64_t grp = (int64_t)blockIdx.x * Geo::GROUPS_PER_WAVE + lane / Geo::LANES;
  const int li = lane % Geo::LANES;
  const bool live = grp < groups;
  const float sc = live ? params[grp] : 1.f, z = live ? params[groups + grp] : 0.f;
  float tsum = 0.f, esum = 0.f;
#pragma unroll
  for (int j = 0; j < Geo::E; ++j) {
    const float wf = live ? h(bf16_lo_to_f32(w[grp * G + li + j * Geo::LANES])) : 0.f;
    const float wq = fminf(fmaxf(rintf(h(h(wf * sc) + z)), 0.f), 15.f);
    const float wr = h(h(wq - z) / sc);
    const float x = h(wf - wr);
    const float ax = fabsf(x);
    const float pw = h(powf(ax, -0.3f));           // |x|^(lp_norm - 1), lp_norm = 0.7
    const float d = h(ax - h(inv_beta * pw));
    const float r = (d != d) ? d : fmaxf(d, 0.f);  // relu keeps NaN
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    const float we = h(sgn * r);
    tsum += h(wq - h(h(wf - we) * sc));            // fp16 values of magnitude <= 16: the fp32 sum is exact in any order
    esum += live ? ax : 0.f;
  }
#pragma unroll
  for (int off = 1; off < Geo::LANES; off <<= 1) tsum += __shfl_xor(tsum, off);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) esum += __shfl_xor(esum, off);
  if (live && li == 0) params[groups + grp] = h(tsum / (float)G);
  if (lane == 0) atomicAdd(err, (double)esum);
}

// codes with the optimised (scale, zero), and the conversion to the tinygemm convention:
//   q = clamp(round(W * scale + zero), 0, 15)   (fp32 weights x fp16 params promote to fp32: two fp32 roundings)
//   scale_out = bf16(fp16(1 / scale));  zero_out = bf16(fp16((8 - zero) * scale_out_fp16))
// bytes [N][K/2] with even k in the HIGH nibble (the input of _convert_weight_to_int4pack), scale_and_zero [K/g][N][2]
template <int G>
__global__ __launch_bounds__(64) void hqq_finish_kernel(const uint16_t* __restrict__ w, const float* __restrict__ params, uint8_t* __restrict__ bytes,
                                                        uint32_t* __restrict__ sz, int64_t groups, int64_t N, int64_t K) {
  const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;  // one thread per byte (two codes of one row)
  if (idx >= N * K / 2) return;
  const int64_t n = idx / (K / 2), k = (idx % (K / 2)) * 2;
  const int64_t grp = (n * K + k) / G;
  const float sc = params[grp], z = params[groups + grp];
  const uint32_t pair = *reinterpret_cast<const uint32_t*>(w + n * K + k);
  const float q0 = fminf(fmaxf(rintf(__fadd_rn(__fmul_rn(bf16_lo_to_f32(pair), sc), z)), 0.f), 15.f);
  const float q1 = fminf(fmaxf(rintf(__fadd_rn(__fmul_rn(bf16_hi_to_f32(pair), sc), z)), 0.f), 15.f);
  bytes[idx] = (uint8_t)(((int)q0 << 4) | (int)q1);
  if (k % G == 0) {
    const float s16 = h(1.0f / sc);
    const float z16 = h(__fmul_rn(8.0f - z, s16));
    sz[(k / G) * N + n] = (uint32_t)f32_to_bf16_bits(s16) | ((uint32_t)f32_to_bf16_bits(z16) << 16);
  }
}

template <int G>
int run_hqq(const uint16_t* w, uint8_t* bytes, uint32_t* sz, float* params, double* err, int64_t N, int64_t K, hipStream_t s) {
  using Geo = HqqGeom<G>;
  const int64_t groups = N * K / G;
  const int64_t blocks = (groups + Geo::GROUPS_PER_WAVE - 1) / Geo::GROUPS_PER_WAVE;
  AO_REQUIRE(blocks < (1ll << 31) && (N * K / 2 + 63) / 64 < (1ll << 31), "ao_int4_quantize_hqq: tensor too large for one launch");
  ao::launch(hqq_init_kernel<G>, dim3((unsigned)blocks), dim3(64), 0, s, w, params, groups);
  double beta = 1e1, best = 1e4;
  const double kappa = 1.01;
  for (int it = 0; it < 20; ++it) {
    hipError_t e = hipMemsetAsync(err, 0, sizeof(double), s);
    if (e != hipSuccess) return hip_failed(e, "hipMemsetAsync(hqq error)");
    ao::launch(hqq_iter_kernel<G>, dim3((unsigned)blocks), dim3(64), 0, s, w, params, err, groups, (float)(1.0 / beta));
    beta *= kappa;
    double sum = 0.0;
    e = hipMemcpyAsync(&sum, err, sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_failed(e, "hqq error read-back");
    const double cur = (double)(float)(_Float16)(float)(sum / (double)(N * K));  // float(fp16 mean)
    if (cur < best) best = cur; else break;  // early_stop
  }
  ao::launch(hqq_finish_kernel<G>, dim3((unsigned)((N * K / 2 + 63) / 64)), dim3(64), 0, s, w, (const float*)params, bytes, sz, groups, N, K);
  AO_LAUNCH_CHECK("hqq kernels launch");
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int64_t ao_int4_hqq_workspace_bytes(int64_t N, int64_t K, int group_size) {
  if (N <= 0 || K <= 0 || group_size <= 0) return -1;
  return 2 * (N * K / group_size) * (int64_t)sizeof(float) + 16;
}

extern "C" int ao_int4_quantize_hqq(const uint16_t* w, uint8_t* nibble_bytes, uint16_t* scale_and_zero, void* workspace, int64_t N, int64_t K,
                                    int group_size, void* stream) {
  AO_REQUIRE(N > 0 && K > 0, "ao_int4_quantize_hqq: bad shape N=%lld K=%lld", (long long)N, (long long)K);
  AO_REQUIRE(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256,
             "ao_int4_quantize_hqq: group_size must be one of 32, 64, 128, 256, got %d", group_size);
  AO_REQUIRE(K % group_size == 0 && K % 2 == 0, "ao_int4_quantize_hqq: K=%lld not divisible by group_size=%d", (long long)K, group_size);
  AO_REQUIRE_PTR(w);
  AO_REQUIRE_PTR(nibble_bytes);
  AO_REQUIRE_PTR(scale_and_zero);
  AO_REQUIRE_PTR(workspace);
  const int64_t groups = N * K / group_size;
  float* params = reinterpret_cast<float*>(workspace);
  double* err = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + ((2 * groups * sizeof(float) + 7) & ~(size_t)7));
  uint32_t* sz = reinterpret_cast<uint32_t*>(scale_and_zero);
  hipStream_t s = (hipStream_t)stream;
  switch (group_size) {
    case 32: return run_hqq<32>(w, nibble_bytes, sz, params, err, N, K, s);
    case 64: return run_hqq<64>(w, nibble_bytes, sz, params, err, N, K, s);
    case 128: return run_hqq<128>(w, nibble_bytes, sz, params, err, N, K, s);
    default: return run_hqq<256>(w, nibble_bytes, sz, params, err, N, K, s);
  }
}
