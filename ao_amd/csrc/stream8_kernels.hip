// Weight-streaming small-M GEMM for 1-byte (OCP e4m3) weights on gfx950:
//   * MXFP8 grouped GEMM (aten::_scaled_grouped_mm as called from
//     torchao/prototype/moe_training/mxfp8_grouped_mm.py:541; numerics of
//     _emulated_mxfp8_scaled_grouped_mm_2d_3d, :959-1023)
//   * float8 rowwise linear at decode batch sizes (aten::_scaled_mm as called from
//     torchao/float8/inference.py:104-123)
// Both are HBM-bound on the weight stream (arithmetic intensity ~2*M flop/B), so
// the structure is the int4 GEMV's: a workgroup owns 16 output features of one
// expert, its waves split K, every wave streams its weight rows straight into
// VGPRs (non-temporal, 128 B per row per step) and feeds
// v_mfma_scale_f32_16x16x128_f8f6f4 -- the E8M0 block scales go into the MFMA as
// operands (one byte per lane per 32-k block), so no cuBLAS-style scale swizzle
// and no separate dequant pass exist.  One barrier for the split-K reduction.
#include "common.h"

namespace ao {
namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct Stream8Args {
  const uint8_t* a;        // [M_total][K] e4m3
  const uint8_t* a_scale;  // MX: [M_total][K/32] e8m0
  const uint8_t* b;        // [E][N][K] e4m3
  const uint8_t* b_scale;  // MX: [E][N][K/32] e8m0
  const float* row_scale;  // rowwise: [M_total] fp32 (scale_a)
  const float* col_scale;  // rowwise: [N] fp32 (scale_b)
  const uint16_t* bias;    // rowwise: [N] bf16 or null
  const int32_t* offs;     // grouped: [E] cumulative row ends; null => one group of M_total rows
  uint16_t* out;           // [M_total][N] bf16
  int M_total, N, K, E;
};

// MX = block-scaled operands (scales from memory); otherwise unit scales + fp32 row/col epilogue.
// MT = m-tiles (16 rows each) handled per pass.
template <bool MX, int MT>
__global__ __launch_bounds__(512) void stream8_kernel(Stream8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [nwaves][MT][256]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int ntile = blockIdx.x;
  const int e = blockIdx.y;
  const int row_begin = (p.offs != nullptr && e > 0) ? p.offs[e - 1] : 0;
  const int row_end = (p.offs != nullptr) ? p.offs[e] : p.M_total;
  const int rows_total = row_end - row_begin;
  if (rows_total <= 0) return;  // uniform: empty group

  const int ksteps = p.K >> 7;  // 128 k per step
  const int ks0 = (ksteps * wave) / nwaves;
  const int ks1 = (ksteps * (wave + 1)) / nwaves;
  const int n = ntile * 16 + (lane & 15);
  const int kq = lane >> 4;
  const int kblocks = p.K >> 5;

  // operand layout of the K=128 scaled MFMA (probed on gfx950, tools/probe_mfma_scale.hip):
  // lane group kq holds k = 16*kq..+15 and 64+16*kq..+15 (two K=64 halves), while its
  // scale byte applies to the 32 consecutive k of block kq.
  const uint8_t* brow = p.b + ((size_t)e * p.N + n) * p.K + kq * 16;
  const uint8_t* bsrow = MX ? p.b_scale + ((size_t)e * p.N + n) * kblocks : nullptr;

  for (int m_base = 0; m_base < rows_total; m_base += 16 * MT) {
    const int rows = min(16 * MT, rows_total - m_base);
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this lane's A rows (one per m-tile), clamped into the group; rows beyond
    // `rows` contribute zeros
    const uint8_t* arow[MT];
    const uint8_t* asrow[MT];
    bool valid[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int r = t * 16 + (lane & 15);
      valid[t] = r < rows;
      const int rr = row_begin + m_base + min(r, rows - 1);
      arow[t] = p.a + (size_t)rr * p.K + kq * 16;
      asrow[t] = MX ? p.a_scale + (size_t)rr * kblocks : nullptr;
    }

    for (int ks = ks0; ks < ks1; ++ks) {
      const u32x4 b0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(brow + (size_t)ks * 128));
      const u32x4 b1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(brow + (size_t)ks * 128 + 64));
      int sb = 127;
      if (MX) sb = (int)((*reinterpret_cast<const uint32_t*>(bsrow + ks * 4)) >> (8 * kq)) & 0xff;
      const i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (t * 16 < rows) {  // uniform
          u32x4 a0 = *reinterpret_cast<const u32x4*>(arow[t] + (size_t)ks * 128);
          u32x4 a1 = *reinterpret_cast<const u32x4*>(arow[t] + (size_t)ks * 128 + 64);
          int sa = 127;
          if (MX) sa = (int)((*reinterpret_cast<const uint32_t*>(asrow[t] + ks * 4)) >> (8 * kq)) & 0xff;
          if (!valid[t]) { a0 = u32x4{0, 0, 0, 0}; a1 = u32x4{0, 0, 0, 0}; sa = 127; }
          const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
          acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc[t], 0, 0, 0, sa, 0, sb);
        }
      }
    }

    // split-K reduction across waves: red[wave][t][row 16][col 16]
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float* r = red + ((size_t)wave * MT + t) * 256 + (kq * 4) * 16 + (lane & 15);
      r[0] = acc[t].x; r[16] = acc[t].y; r[32] = acc[t].z; r[48] = acc[t].w;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * 256; idx += blockDim.x) {
      const int t = idx >> 8, rc = idx & 255;
      const int row = t * 16 + (rc >> 4), col = rc & 15;
      if (row < rows) {
        float sum = 0.f;
        for (int w = 0; w < nwaves; ++w) sum += red[((size_t)w * MT + t) * 256 + rc];
        const int gm = row_begin + m_base + row, gn = ntile * 16 + col;
        if (!MX) {
          sum = sum * p.row_scale[gm] * p.col_scale[gn];
          if (p.bias != nullptr) sum += bf16_lo_to_f32(p.bias[gn]);
        }
        p.out[(size_t)gm * p.N + gn] = f32_to_bf16_bits(sum);
      }
    }
    __syncthreads();
  }
}

template <bool MX>
int launch_stream8(const Stream8Args& p, int max_rows_per_group, hipStream_t stream) {
  const int ksteps = p.K >> 7;
  int wpb = 1;
  while (wpb < 8 && ksteps / (wpb * 2) >= 2) wpb *= 2;
  dim3 grid((unsigned)(p.N / 16), (unsigned)p.E), block(wpb * 64);
  const int mt = max_rows_per_group <= 16 ? 1 : (max_rows_per_group <= 32 ? 2 : 4);
  const size_t smem = (size_t)wpb * mt * 256 * sizeof(float);
  switch (mt) {
    case 1: ao::launch(stream8_kernel<MX, 1>, grid, block, smem, stream, p); break;
    case 2: ao::launch(stream8_kernel<MX, 2>, grid, block, smem, stream, p); break;
    default: ao::launch(stream8_kernel<MX, 4>, grid, block, smem, stream, p); break;
  }
  AO_LAUNCH_CHECK("stream8_kernel launch");
  return AO_OK;
}

}  // namespace

// used by gemm8_kernels.hip for the small-M float8 rowwise case
int fp8_rowwise_stream(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b,
                       const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  Stream8Args p{};
  p.a = a; p.b = b; p.row_scale = scale_a; p.col_scale = scale_b; p.bias = bias; p.out = y;
  p.M_total = (int)M; p.N = (int)N; p.K = (int)K; p.E = 1;
  return launch_stream8<false>(p, (int)M, stream);
}

}  // namespace ao

using namespace ao;

extern "C" int ao_mxfp8_grouped_mm(const uint8_t* a, const uint8_t* a_scale, const uint8_t* b,
                                   const uint8_t* b_scale, const int32_t* offs, uint16_t* out, int64_t M_total,
                                   int64_t N, int64_t K, int64_t E, void* stream) {
  AO_REQUIRE(M_total >= 0 && N > 0 && K > 0 && E > 0, "ao_mxfp8_grouped_mm: bad shape M_total=%lld N=%lld K=%lld E=%lld",
             (long long)M_total, (long long)N, (long long)K, (long long)E);
  AO_REQUIRE(K % 128 == 0, "ao_mxfp8_grouped_mm: K=%lld must be a multiple of 128", (long long)K);
  AO_REQUIRE(N % 16 == 0, "ao_mxfp8_grouped_mm: N=%lld must be a multiple of 16", (long long)N);
  AO_REQUIRE(M_total < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && E < 65536,
             "ao_mxfp8_grouped_mm: dimension too large");
  if (M_total == 0) return AO_OK;
  AO_REQUIRE_PTR(a);
  AO_REQUIRE_PTR(a_scale);
  AO_REQUIRE_PTR(b);
  AO_REQUIRE_PTR(b_scale);
  AO_REQUIRE_PTR(out);
  AO_REQUIRE(offs != nullptr || E == 1, "ao_mxfp8_grouped_mm: offs is required when E > 1");
  Stream8Args p{};
  p.a = a; p.a_scale = a_scale; p.b = b; p.b_scale = b_scale; p.offs = offs; p.out = out;
  p.M_total = (int)M_total; p.N = (int)N; p.K = (int)K; p.E = (int)E;
  // group sizes live on the device; size the m-tiling for the worst case the
  // caller can have (a group cannot exceed M_total rows)
  return launch_stream8<true>(p, (int)M_total, (hipStream_t)stream);
}
