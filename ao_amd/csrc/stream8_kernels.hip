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

#include <type_traits>

namespace ao {
// rb8_kernels.hip: LDS-staged weight-streaming form (full-line weight requests)
int fp8_rowwise_grouped_rb(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, const int32_t* offs, uint16_t* out,
                           int64_t M_total, int64_t N, int64_t K, int64_t E, hipStream_t stream);  // rb8_kernels.hip
int mxfp8_grouped_rb(const uint8_t* a, const uint8_t* a_scale, const uint8_t* b, const uint8_t* b_scale, const int32_t* offs, uint16_t* out,
                     int64_t M_total, int64_t N, int64_t K, int64_t E, int64_t rows_hint, hipStream_t stream);
bool mxfp8_grouped_dyn_fits(int64_t M_total, int64_t N, int64_t K, int64_t E, bool have_offs, int products);
int mxfp8_grouped_stream16(const void* a, const uint8_t* a_scale, const uint8_t* b, const uint8_t* b_scale, const uint8_t* b2, const uint8_t* b2_scale,
                           const int32_t* offs, uint16_t* out, uint16_t* out2, int64_t M_total, int64_t N, int64_t K, int64_t E, int scaling_mode,
                           hipStream_t stream);
thread_local int g_mx_variant = 0;  // profiling / A-B tests (ao_gemm8_set_variant 110 / 111): 0 by shape, 1 always the LDS-staged kernel, 2 never
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

// KIND: block-scaled e4m3 operands (scales from memory) / e4m3 with unit block scales + fp32 row/col epilogue / int8 with
// int32 accumulation (two v_mfma_i32_16x16x64_i8 per step) + the int8 linear's two-rounding epilogue.
// MT = m-tiles (16 rows each) handled per pass.
enum Stream8Kind { S8_FP8_ROWWISE = 0, S8_MX = 1, S8_INT8 = 2 };
template <int KIND, int MT>
__global__ __launch_bounds__(512) void stream8_kernel(Stream8Args p) {
  constexpr bool MX = (KIND == S8_MX), INT8 = (KIND == S8_INT8);
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
          if constexpr (INT8) {  // acc holds int32 bit patterns
            i32x4 c = __builtin_bit_cast(i32x4, acc[t]);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, b1), c, 0, 0, 0);
            acc[t] = __builtin_bit_cast(f32x4, c);
          } else {
            const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
            acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc[t], 0, 0, 0, sa, 0, sb);
          }
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
        int isum = 0;
        for (int w = 0; w < nwaves; ++w) {
          if constexpr (INT8) isum += __float_as_int(red[((size_t)w * MT + t) * 256 + rc]);
          else sum += red[((size_t)w * MT + t) * 256 + rc];
        }
        const int gm = row_begin + m_base + row, gn = ntile * 16 + col;
        if constexpr (INT8) {
          // t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))   (int8_tensor.py:315-359)
          sum = round_bf16((float)isum * p.row_scale[gm]) * p.col_scale[gn];
          if (p.bias != nullptr) sum += bf16_lo_to_f32(p.bias[gn]);
        } else if (!MX) {
          sum = sum * p.row_scale[gm] * p.col_scale[gn];
          if (p.bias != nullptr) sum += bf16_lo_to_f32(p.bias[gn]);
        }
        p.out[(size_t)gm * p.N + gn] = f32_to_bf16_bits(sum);
      }
    }
    __syncthreads();
  }
}

// Buffer addressing: SGPR descriptor + SGPR offset + ONE 32-bit VGPR offset per lane, so a dozen
// streams (TN weight tiles, MT activation rows, their scales) cost no 64-bit VGPR pointers; reads
// past num_records return 0 (tiles past N need no clamp).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr int kAuxNT = 2;  // non-temporal: streamed once
template <int AUX>
__device__ __forceinline__ u32x4 buf_load_b128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
template <int AUX>
__device__ __forceinline__ uint32_t buf_load_b32(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, AUX);
}

// ---------------------------------------------------------------------------------------------
// MXFP8 grouped GEMM, A-stationary form (the MoE decode/prefill-at-small-batch case of BASELINE
// config 5: a few dozen token rows per expert against 58.7 MB of expert weights).
//
// stream8_kernel above re-reads the expert's A rows [rows][K] from L2 for EVERY 16-wide n-tile
// (2-4x the weight bytes per workgroup): it runs at 2.2 TB/s.  Here a workgroup owns TN n-tiles of
// one expert, its 8 waves split each 2048-k chunk (2 k-steps of 128 per wave), a wave loads its A
// fragments for the chunk ONCE into registers and streams the TN weight tiles past them through a
// 4-deep register ring that runs on across chunks; per-tile accumulators stay in registers for the
// whole K loop and meet the other waves' once, at the end (LDS, rounds of 4 tiles).
// ---------------------------------------------------------------------------------------------
template <int MT, int TN>
__global__ __launch_bounds__(256) void mx_grouped_kernel(Stream8Args p) {
  constexpr int W = 4, S = 2;                          // waves, k-steps per wave per chunk (4-wave workgroups: three fit a CU at ~150 VGPRs, so one's prologue/reduction overlaps the others' streaming)
  constexpr int R = TN < 4 ? TN : 4;                   // tiles per reduction round
  constexpr int D = 2;                                 // ring depth in tiles (18 VGPRs per stage; the A double buffer takes the rest)
  static_assert(TN % D == 0 && TN % R == 0, "ring and reduction rounds must divide the tile group");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [W][R][MT][256]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = blockIdx.y;
  const int row_begin = (e > 0) ? p.offs[e - 1] : 0;
  const int rows_total = p.offs[e] - row_begin;
  if (rows_total <= 0) return;  // uniform: empty group

  const int ntiles = p.N >> 4;
  const int tile0 = blockIdx.x * TN;
  const int kq = lane >> 4, nl = lane & 15;
  const int kblocks = p.K >> 5;
  const int chunks = p.K / (W * S * 128);  // 1024 k per chunk

  // this expert's weights and scales as buffers; lane offset inside a 16-row tile
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.b + (size_t)e * p.N * p.K, (uint32_t)((size_t)p.N * p.K));
  const __amdgpu_buffer_rsrc_t rbs = make_rsrc(p.b_scale + (size_t)e * p.N * kblocks, (uint32_t)((size_t)p.N * kblocks));
  const uint32_t b_off = (uint32_t)nl * (uint32_t)p.K + kq * 16;
  const uint32_t bs_off = (uint32_t)nl * (uint32_t)kblocks;

  for (int m_base = 0; m_base < rows_total; m_base += 16 * MT) {
    const int rows = min(16 * MT, rows_total - m_base);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.a + (size_t)(row_begin + m_base) * p.K, (uint32_t)((size_t)rows * p.K));
    const __amdgpu_buffer_rsrc_t ras = make_rsrc(p.a_scale + (size_t)(row_begin + m_base) * kblocks, (uint32_t)((size_t)rows * kblocks));
    uint32_t a_off[MT], as_off[MT];
    bool valid[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int r = t * 16 + nl;
      valid[t] = r < rows;
      const uint32_t rr = (uint32_t)min(r, rows - 1);
      a_off[t] = rr * (uint32_t)p.K + kq * 16;
      as_off[t] = rr * (uint32_t)kblocks;
    }
    f32x4 acc[TN][MT];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Stage {
      u32x4 b[S][2];
      uint32_t sb[S];
    };
    Stage st[D];
    // k-step s of this wave in chunk c
    auto kstep = [&](int c, int s) { return c * (W * S) + wave * S + s; };
    auto issue = [&](Stage& sg, int j, int c) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int ks = kstep(c, s);
        const uint32_t trow = (uint32_t)(tile0 + j) * 16u;  // tiles past N read zeros (buffer bounds)
        sg.b[s][0] = buf_load_b128<kAuxNT>(rb, b_off, trow * (uint32_t)p.K + ks * 128);
        sg.b[s][1] = buf_load_b128<kAuxNT>(rb, b_off + 64, trow * (uint32_t)p.K + ks * 128);
        sg.sb[s] = buf_load_b32<0>(rbs, bs_off, trow * (uint32_t)kblocks + ks * 4);
      }
    };
    // this wave's A fragments of a chunk (L2-resident).  Loads return in order, so they must be OLDER
    // than the weight stages whose MFMAs need them: chunk 0's go out before the ring prologue, chunk
    // c+1's at the top of chunk c (double buffer) -- fetching them at the top of their own chunk
    // meant waiting behind every weight stage in flight, i.e. draining the ring once per chunk.
    struct AFrag {
      u32x4 a[S][MT][2];
      uint32_t sa[S][MT];
    };
    auto load_a = [&](AFrag& f, int c) {
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int ks = kstep(c, s);
          f.a[s][t][0] = buf_load_b128<0>(ra, a_off[t], ks * 128);
          f.a[s][t][1] = buf_load_b128<0>(ra, a_off[t] + 64, ks * 128);
          f.sa[s][t] = buf_load_b32<0>(ras, as_off[t], ks * 4);
        }
    };
    AFrag cur, nxt;
    load_a(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ring prologue: tiles 0..D-1 of chunk 0
#pragma unroll
    for (int d = 0; d < D; ++d) issue(st[d], d, 0);

    // One chunk: build the MFMA operands from `cur`, prefetch the next chunk's A, stream the TN tiles.
    // No branch inside (rows beyond the group carry zero A fragments instead of skipping their MFMAs,
    // the last chunk is a separate instantiation): any wave-uniform branch between a load and its use
    // makes the compiler wait vmcnt(0), which drains the ring (11 GB/s per CU instead of 25).
    auto chunk_body = [&](int c, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      i32x8 af[S][MT];
      int sa[S][MT];
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          u32x4 a0 = cur.a[s][t][0], a1 = cur.a[s][t][1];
          int sc = (int)(cur.sa[s][t] >> (8 * kq)) & 0xff;
          if (!valid[t]) { a0 = u32x4{0, 0, 0, 0}; a1 = u32x4{0, 0, 0, 0}; sc = 127; }  // v_cndmask, not a branch
          af[s][t] = i32x8{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
          sa[s][t] = sc;
        }
      if (!LAST) {
        load_a(nxt, c + 1);
        __builtin_amdgcn_sched_barrier(0);  // keep these loads OLDER than the weight stages issued below
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const Stage& sg = st[j % D];
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int sb = (int)(sg.sb[s] >> (8 * kq)) & 0xff;
          const i32x8 bf = {(int)sg.b[s][0].x, (int)sg.b[s][0].y, (int)sg.b[s][0].z, (int)sg.b[s][0].w,
                            (int)sg.b[s][1].x, (int)sg.b[s][1].y, (int)sg.b[s][1].z, (int)sg.b[s][1].w};
#pragma unroll
          for (int t = 0; t < MT; ++t)
            acc[j][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[s][t], bf, acc[j][t], 0, 0, 0, sa[s][t], 0, sb);
        }
        if (!LAST) issue(st[j % D], (j + D) % TN, c + (j + D) / TN);
        else if (j + D < TN) issue(st[j % D], j + D, c);
      }
      if (!LAST) cur = nxt;
    };
    for (int c = 0; c + 1 < chunks; ++c) chunk_body(c, std::false_type{});
    chunk_body(chunks - 1, std::true_type{});

    // cross-wave (split-K) reduction, R tiles per round: red[wave][r][t][row 16][col 16]
#pragma unroll
    for (int j0 = 0; j0 < TN; j0 += R) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          float* q = red + (((size_t)wave * R + r) * MT + t) * 256 + (kq * 4) * 16 + nl;
          q[0] = acc[j0 + r][t].x; q[16] = acc[j0 + r][t].y; q[32] = acc[j0 + r][t].z; q[48] = acc[j0 + r][t].w;
        }
      __syncthreads();
      for (int idx = threadIdx.x; idx < R * MT * 256; idx += W * 64) {
        const int r = idx / (MT * 256), t = (idx / 256) % MT, rc = idx & 255;
        const int row = t * 16 + (rc >> 4), col = rc & 15;
        const int tile = tile0 + j0 + r;
        if (row < rows && tile < ntiles) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < W; ++w) sum += red[(((size_t)w * R + r) * MT + t) * 256 + rc];
          p.out[(size_t)(row_begin + m_base + row) * p.N + tile * 16 + col] = f32_to_bf16_bits(sum);
        }
      }
      __syncthreads();
    }
  }
}

template <int MT, int TN>
int launch_mx_grouped_tn(const Stream8Args& p, hipStream_t stream) {
  const int ntiles = p.N >> 4;
  constexpr int R = TN < 4 ? TN : 4;
  const size_t smem = (size_t)4 * R * MT * 256 * sizeof(float);
  auto kern = mx_grouped_kernel<MT, TN>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_failed(e, "hipFuncSetAttribute(mx_grouped_kernel)");
  }
  ao::launch(kern, dim3((unsigned)((ntiles + TN - 1) / TN), (unsigned)p.E), dim3(256), smem, stream, p);
  AO_LAUNCH_CHECK("mx_grouped_kernel launch");
  return AO_OK;
}

// n-tiles per workgroup: as many as the accumulators allow (A is fetched once per workgroup) while
// the grid still has >= 1024 workgroups -- only experts that received tokens do any work
template <int MT>
int launch_mx_grouped(const Stream8Args& p, hipStream_t stream) {
  const int64_t ntiles = p.N >> 4;
  auto enough = [&](int tn) { return ((ntiles + tn - 1) / tn) * p.E >= 1024; };
  if (MT == 1 && enough(8)) return launch_mx_grouped_tn<1, 8>(p, stream);
  if (enough(4)) return launch_mx_grouped_tn<MT, 4>(p, stream);
  return launch_mx_grouped_tn<MT, 2>(p, stream);
}

template <int KIND>
int launch_stream8(const Stream8Args& p, int max_rows_per_group, hipStream_t stream) {
  const int ksteps = p.K >> 7;
  int wpb = 1;
  while (wpb < 8 && ksteps / (wpb * 2) >= 2) wpb *= 2;
  dim3 grid((unsigned)(p.N / 16), (unsigned)p.E), block(wpb * 64);
  const int mt = max_rows_per_group <= 16 ? 1 : (max_rows_per_group <= 32 ? 2 : 4);
  const size_t smem = (size_t)wpb * mt * 256 * sizeof(float);
  switch (mt) {
    case 1: ao::launch(stream8_kernel<KIND, 1>, grid, block, smem, stream, p); break;
    case 2: ao::launch(stream8_kernel<KIND, 2>, grid, block, smem, stream, p); break;
    default: ao::launch(stream8_kernel<KIND, 4>, grid, block, smem, stream, p); break;
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
  return launch_stream8<S8_FP8_ROWWISE>(p, (int)M, stream);
}

// the same for the int8 dynamic-activation linear at decode batch sizes (int8_tensor.py:305-359)
int int8_scaled_stream(const int8_t* a, const int8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                       int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  Stream8Args p{};
  p.a = reinterpret_cast<const uint8_t*>(a); p.b = reinterpret_cast<const uint8_t*>(b);
  p.row_scale = scale_a; p.col_scale = scale_b; p.bias = bias; p.out = y;
  p.M_total = (int)M; p.N = (int)N; p.K = (int)K; p.E = 1;
  return launch_stream8<S8_INT8>(p, (int)M, stream);
}

}  // namespace ao

using namespace ao;

extern "C" int ao_fp8_grouped_mm(const uint8_t* a, const float* scale_a, const uint8_t* b, const float* scale_b, const int32_t* offs,
                                 uint16_t* out, int64_t M_total, int64_t N, int64_t K, int64_t E, void* stream) {
  AO_REQUIRE(M_total >= 0 && N > 0 && K > 0 && E > 0, "ao_fp8_grouped_mm: bad shape M_total=%lld N=%lld K=%lld E=%lld", (long long)M_total,
             (long long)N, (long long)K, (long long)E);
  AO_REQUIRE(K % 128 == 0 && N % 16 == 0, "ao_fp8_grouped_mm: K=%lld must be a multiple of 128 and N=%lld of 16", (long long)K, (long long)N);
  const int64_t bm = (M_total <= 48 * E) ? 64 : 128;
  AO_REQUIRE(M_total * K < (1ll << 32) && N * K < (1ll << 32) && E + (M_total + bm - 1) / bm <= 65535,
             "ao_fp8_grouped_mm: tensor too large for one launch (M_total * K and N * K must stay below 4 Gi elements)");
  if (M_total == 0) return AO_OK;
  AO_REQUIRE_PTR(a);
  AO_REQUIRE_PTR(scale_a);
  AO_REQUIRE_PTR(b);
  AO_REQUIRE_PTR(scale_b);
  AO_REQUIRE_PTR(offs);
  AO_REQUIRE_PTR(out);
  return fp8_rowwise_grouped_rb(a, b, scale_a, scale_b, offs, out, M_total, N, K, E, (hipStream_t)stream);
}

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
  // The LDS-staged weight-streaming kernel (rb8_kernels.hip) is the product path; its slab height follows the average group
  // size.  Mixtral-8x7B expert shapes, 8 experts (w1, w2; us): 16 rows per expert -- A-stationary kernel below 97 / 120,
  // LDS-staged 89 / 110; 128 rows per expert -- 381 / 474 (the A-stationary kernel re-streams the weights per 64-row pass) vs
  // 211 / 172.  Variant 111 keeps the older kernels reachable for A/B runs.
  // (it addresses activation rows with 32-bit byte offsets and puts experts x slabs on grid.y: tensors beyond either bound --
  // M_total * K >= 4 GiB, or more than 65535 possible non-empty (expert, slab) pairs -- take the per-tile kernels below, which have neither limit)
  const int64_t rb_bm = (M_total <= 48 * (offs != nullptr ? E : 1)) ? 64 : 128;
  const bool rb_ok = M_total * K < (1ll << 32) && N * K < (1ll << 32) && (offs != nullptr ? E : 0) + (M_total + rb_bm - 1) / rb_bm <= 65535;
  if (g_mx_variant != 2 && rb_ok) return mxfp8_grouped_rb(a, a_scale, b, b_scale, offs, out, M_total, N, K, E, M_total, (hipStream_t)stream);
  if (offs != nullptr && K % 2048 == 0) {
    // Group sizes live on the device.  Size the m-tiling for twice the AVERAGE group: a larger group
    // takes another pass over its expert's weights (correct, slower), while sizing for the worst
    // case (one group of M_total rows) would carry 4 m-tiles of A and accumulators everywhere.
    const int64_t guess = (2 * M_total + E - 1) / E;
    if (guess <= 16) return launch_mx_grouped<1>(p, (hipStream_t)stream);
    if (guess <= 32) return launch_mx_grouped<2>(p, (hipStream_t)stream);
    return launch_mx_grouped<4>(p, (hipStream_t)stream);
  }
  // other K: the per-tile kernel, m-tiling for the worst case (a group cannot exceed M_total rows)
  return launch_stream8<S8_MX>(p, (int)M_total, (hipStream_t)stream);
}

// Host-only: whether ao_mxfp8_grouped_mm_dyn (products = 1) / the pair forms (products = 2) take the shape: decode-size groups
// (M_total <= 48 per group, <= 64 groups), K % 512 == 0, N % 16 == 0.
extern "C" int ao_mxfp8_grouped_mm_dyn_fits(int64_t M_total, int64_t N, int64_t K, int64_t E) {
  return mxfp8_grouped_dyn_fits(M_total, N, K, E, true, 1) ? 1 : 0;
}
extern "C" int ao_mxfp8_grouped_mm_pair_fits(int64_t M_total, int64_t N, int64_t K, int64_t E) {
  return mxfp8_grouped_dyn_fits(M_total, N, K, E, true, 2) ? 1 : 0;
}

namespace {
int mx_stream16_entry(const char* fn, const void* a, const uint8_t* a_scale, const uint8_t* b, const uint8_t* b_scale, const uint8_t* b2,
                      const uint8_t* b2_scale, const int32_t* offs, uint16_t* out, uint16_t* out2, int64_t M_total, int64_t N, int64_t K, int64_t E,
                      int scaling_mode, bool pair, void* stream) {
  AO_REQUIRE(a_scale != nullptr || scaling_mode == 0 || scaling_mode == 1, "%s: scaling_mode must be AO_MX_SCALE_FLOOR or AO_MX_SCALE_RCEIL, got %d", fn, scaling_mode);
  AO_REQUIRE(M_total >= 0 && N > 0 && K > 0 && E > 0, "%s: bad shape M_total=%lld N=%lld K=%lld E=%lld", fn, (long long)M_total, (long long)N, (long long)K, (long long)E);
  if (M_total == 0) return AO_OK;
  AO_REQUIRE_PTR(a);
  AO_REQUIRE_PTR(b);
  AO_REQUIRE_PTR(b_scale);
  AO_REQUIRE_PTR(offs);
  AO_REQUIRE_PTR(out);
  if (pair) {
    AO_REQUIRE_PTR(b2);
    AO_REQUIRE_PTR(b2_scale);
    AO_REQUIRE_PTR(out2);
  }
  const bool aligned = ((uintptr_t)b_scale % 16 == 0) && ((uintptr_t)a % 16 == 0) && (a_scale == nullptr || (uintptr_t)a_scale % 16 == 0) &&
                       (!pair || (uintptr_t)b2_scale % 16 == 0);
  AO_REQUIRE(mxfp8_grouped_dyn_fits(M_total, N, K, E, true, pair ? 2 : 1) && aligned,
             "%s: shape M_total=%lld N=%lld K=%lld E=%lld is not a decode-size grouped product (ao_mxfp8_grouped_mm_dyn_fits / _pair_fits; scales and "
             "activations 16-byte aligned): cast with ao_mxfp8_quantize_rowwise and call ao_mxfp8_grouped_mm per weight", fn, (long long)M_total, (long long)N,
             (long long)K, (long long)E);
  return mxfp8_grouped_stream16(a, a_scale, b, b_scale, pair ? b2 : nullptr, pair ? b2_scale : nullptr, offs, out, pair ? out2 : nullptr, M_total, N, K, E,
                                scaling_mode, (hipStream_t)stream);
}
}  // namespace

// _to_mxfp8_then_scaled_grouped_mm's forward in ONE launch (mxfp8_grouped_mm.py:330-371: to_mx(A, block 32, scaling_mode) then the 2d-3d
// grouped mm): a is the BF16 activation matrix, the cast happens in the kernel's A-fill with the stand-alone cast's arithmetic.
extern "C" int ao_mxfp8_grouped_mm_dyn(const uint16_t* a, const uint8_t* b, const uint8_t* b_scale, const int32_t* offs, uint16_t* out,
                                       int64_t M_total, int64_t N, int64_t K, int64_t E, int scaling_mode, void* stream) {
  return mx_stream16_entry(__func__, a, nullptr, b, b_scale, nullptr, nullptr, offs, out, nullptr, M_total, N, K, E, scaling_mode, false, stream);
}
// Two expert-weight tensors of ONE shape against the same activations in one launch -- an MoE layer's w1 and w3 (x @ w1, x @ w3: the
// reference calls _to_mxfp8_then_scaled_grouped_mm once per weight, casting x twice): out1 / out3 are bit-identical to two single calls.
extern "C" int ao_mxfp8_grouped_mm_dyn_pair(const uint16_t* a, const uint8_t* b1, const uint8_t* b1_scale, const uint8_t* b3, const uint8_t* b3_scale,
                                            const int32_t* offs, uint16_t* out1, uint16_t* out3, int64_t M_total, int64_t N, int64_t K, int64_t E,
                                            int scaling_mode, void* stream) {
  return mx_stream16_entry(__func__, a, nullptr, b1, b1_scale, b3, b3_scale, offs, out1, out3, M_total, N, K, E, scaling_mode, true, stream);
}
// The same with activations the caller cast already (e4m3 codes + E8M0 scales: the output of the EP dispatch, or a cast shared with other consumers).
extern "C" int ao_mxfp8_grouped_mm_pair(const uint8_t* a, const uint8_t* a_scale, const uint8_t* b1, const uint8_t* b1_scale, const uint8_t* b3,
                                        const uint8_t* b3_scale, const int32_t* offs, uint16_t* out1, uint16_t* out3, int64_t M_total, int64_t N,
                                        int64_t K, int64_t E, void* stream) {
  AO_REQUIRE_PTR(a_scale);
  return mx_stream16_entry(__func__, a, a_scale, b1, b1_scale, b3, b3_scale, offs, out1, out3, M_total, N, K, E, 0, true, stream);
}
