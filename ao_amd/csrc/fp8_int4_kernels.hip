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
#include "quant_math.h"

namespace ao {
namespace {

typedef long i64_t;
thread_local int g_fp8_int4_mt = 0;  // A/B: m-tiles per workgroup forced (1, 2, 4); 0 = by M
thread_local bool g_fp8_int4_nt1 = false;  // A/B: one n-tile per workgroup

// FUSE (round 4, SURVEY 8 f1 for this path): the per-row e4m3 cast of the activation inside the launch -- x arrives as bf16 [M][K]
// (M <= 16, M (K + 16) <= 64 KiB), is cast ONCE per workgroup into an LDS copy of the codes with quant_math.h's arithmetic (the bits of
// ao_fp8_quantize_rowwise), and the waves take their blocks' x slices from there instead of from global memory.
//   1: M == 1 -- a wave casts only the k-run it multiplies itself, held in registers between amax and cast; its loads go out before the
//      ring's, the one workgroup barrier is the amax exchange;
//   2: 2 <= M <= 16 -- the workgroup casts the whole activation (two passes over the L2-resident rows), then the ring is requested.
// Without it the stand-alone cast cost this path 28 % (818 -> 587 tok/s on the Llama-3-8B linears, profiles/fp8_int4_r03.jsonl).
// MT (round 5): 16-row m-tiles per workgroup, 1, 2 or 4 (FUSE == 0 only).  Rounds 3-4 ran one workgroup per 16 rows, so a batch of M rows
// streamed, masked and scale-decoded every packed block ceil(M / 16) times; with MT m-tiles a block's B operands (the two mask
// operations per word) and its (scale, zero) pair are built once and multiplied against MT activation tiles: MT x fewer weight
// requests and nibble expansions per output.
// NT (round 5): 16-wide n-tiles per workgroup, 1 or 2.  The staged activation tiles, their byte interleave and the ones-operand MFMA that
// yields sum_k xq per group do not depend on n: with two n-tiles per workgroup they are done once per 32 output columns (12 MFMAs per
// block and m-tile instead of 16, half the LDS staging).
template <int G, int DEPTH, int FUSE = 0, int MT = 1, int NT = 1>
__global__ __launch_bounds__(512) void fp8_int4_mm_kernel(const uint8_t* __restrict__ xq, const float* __restrict__ x_scale,
                                                          const u32x4* __restrict__ qdata, const uint32_t* __restrict__ sz,
                                                          const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int M, int N, int K) {
  constexpr int NG = (G >= 128) ? 1 : (128 / G);  // groups per 128-k block
  constexpr int ROWSTRIDE = 128 + 16;             // bytes per staged x row ([kq][j] pairs of dwords), padded vs bank conflicts
  constexpr int SLAB16 = 16 * ROWSTRIDE;           // one m-tile's rows
  constexpr int SLAB = MT * SLAB16;
  static_assert(FUSE == 0 || (MT == 1 && NT == 1), "the fused cast holds at most 16 rows, one n-tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int ntile = blockIdx.x * NT;  // (first of NT)
  const int m0 = blockIdx.y * (16 * MT);
  const int rows = min(16 * MT, M - m0);
  const int kblocks = K >> 7;
  const int kb0 = (kblocks * wave) / nwaves;
  const int kb1 = (kblocks * (wave + 1)) / nwaves;
  char* slab = smem + wave * SLAB;
  float* red = reinterpret_cast<float*>(smem + nwaves * SLAB);
  // FUSE: [M][K + 16] codes | [nwaves][16] row maxima | [16] row scales, behind the reduction area
  const int cstride = K + 16;
  char* codes = smem + nwaves * (SLAB + MT * NT * 1024);
  float* wmax = reinterpret_cast<float*>(codes + ((M * cstride + 15) & ~15));
  float* rs = wmax + nwaves * 16;

  const int n = ntile * 16 + (lane & 15);  // (+ 16 nt)
  const int kq = lane >> 4;
  const u32x4* wp = qdata + (size_t)ntile * kblocks * 64 + lane;  // (+ nt * kblocks * 64)
  // x slice of a block: lane (r = lane >> 2, j = lane & 3) loads the 32 bytes x[m0 + r][kb * 128 + 32 j ..] (rows past M: the last row)
  const uint8_t* xp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xp[mt] = xq + (size_t)(m0 + min(16 * mt + (lane >> 2), rows - 1)) * K + (lane & 3) * 32;
  // ... and stores the dword pairs (d[kq'], d[4 + kq']) at [r][kq'][j]; lane (m, kq) reads its four pairs (j = 0..3) as 32 contiguous bytes
  char* st_base = slab + (lane >> 2) * ROWSTRIDE + (lane & 3) * 8;
  const char* a_base = slab + (lane & 15) * ROWSTRIDE + kq * 32;

  struct Stage {
    u32x4 w[NT];
    uint32_t sz[NT][NG];
    u32x4 x0[MT], x1[MT];
  };
  Stage st[DEPTH];
  auto issue = [&](Stage& s, int kb) {
    const int kg0 = (G >= 128) ? ((kb * 128) / G) : (kb * NG);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s.w[nt] = __builtin_nontemporal_load(wp + (size_t)nt * kblocks * 64 + (size_t)kb * 64);
#pragma unroll
      for (int i = 0; i < NG; ++i) s.sz[nt][i] = sz[(size_t)(kg0 + i) * N + n + 16 * nt];
    }
    if constexpr (FUSE == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const u32x4* xs = reinterpret_cast<const u32x4*>(xp[mt] + (size_t)kb * 128);
        s.x0[mt] = xs[0];
        s.x1[mt] = xs[1];
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const i64_t ones = 0x3838383838383838L;  // eight e4m3 1.0
  // FUSE: lane (r, j)'s 32 bytes of the block come from the LDS copy of the codes (rows past M: the last row)
  const char* cp = codes + min(lane >> 2, rows - 1) * cstride + (lane & 3) * 32;
  auto consume = [&](const Stage& s, int kb) {
    // stage x: d0..d3 = tile 2j (kq 0..3), d4..d7 = tile 2j + 1 -- every m-tile into its own 16-row slab (wave-private LDS: DS operations
    // of a wave complete in order, no barrier)
    uint32_t xw[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      u32x4 x0, x1;
      if constexpr (FUSE == 0) { x0 = s.x0[mt]; x1 = s.x1[mt]; }
      else { x0 = *reinterpret_cast<const u32x4*>(cp + kb * 128); x1 = *reinterpret_cast<const u32x4*>(cp + kb * 128 + 16); }
      const uint32_t d[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2*>(st_base + mt * SLAB16 + q * 32) = u32x2{d[q], d[4 + q]};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(a_base + mt * SLAB16);       // (A0, B0, A1, B1)
      const u32x4 xb = *reinterpret_cast<const u32x4*>(a_base + mt * SLAB16 + 16);  // (A2, B2, A3, B3)
      xw[mt][0] = xa.x; xw[mt][1] = xa.y; xw[mt][2] = xa.z; xw[mt][3] = xa.w;
      xw[mt][4] = xb.x; xw[mt][5] = xb.y; xw[mt][6] = xb.z; xw[mt][7] = xb.w;
    }
    f32x4 pq[MT][NT], px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      px[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pq[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // B operand: bytes (v0, v4, v1, v5 | v2, v6, v3, v7) = k (t0, t1, t0 + 1, t1 + 1 | t0 + 2, t1 + 2, t0 + 3, t1 + 3)
      i64_t b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint32_t wd = (j == 0) ? s.w[nt].x : (j == 1) ? s.w[nt].y : (j == 2) ? s.w[nt].z : s.w[nt].w;
        const uint32_t lo = wd & 0x0F0F0F0Fu, hi = (wd >> 4) & 0x0F0F0F0Fu;
        b[nt] = (i64_t)(((unsigned long)hi << 32) | lo);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // A operand in the same order: interleave the bytes of A_j = x[.., t0 ..+3] and B_j = x[.., t1 ..+3]
        const uint32_t alo = __builtin_amdgcn_perm(xw[mt][2 * j + 1], xw[mt][2 * j], 0x05010400u);
        const uint32_t ahi = __builtin_amdgcn_perm(xw[mt][2 * j + 1], xw[mt][2 * j], 0x07030602u);
        const i64_t a = (i64_t)(((unsigned long)ahi << 32) | alo);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pq[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b[nt], pq[mt][nt], 0, 0, 0);
        px[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, ones, px[mt], 0, 0, 0);
      }
      if ((j + 1) * 32 % G == 0 || j == 3) {  // a group ends here (G >= 128: once per block)
        const int gi = (G >= 128) ? 0 : ((j * 32) / G);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float sc = bf16_lo_to_f32(s.sz[nt][gi]), zp = bf16_hi_to_f32(s.sz[nt][gi]);
          const float c0 = 512.0f * sc, c1 = zp - 8.0f * sc;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f32x4& a4 = acc[mt][nt];
            const f32x4 q4 = pq[mt][nt], x4 = px[mt];
            a4.x = fmaf(c0, q4.x, fmaf(c1, x4.x, a4.x)); a4.y = fmaf(c0, q4.y, fmaf(c1, x4.y, a4.y));
            a4.z = fmaf(c0, q4.z, fmaf(c1, x4.z, a4.z)); a4.w = fmaf(c0, q4.w, fmaf(c1, x4.w, a4.w));
            pq[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) px[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  const int kb_last = max(kb1 - 1, kb0);
  auto prime = [&] {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(st[d], min(kb0 + d, kb_last));
    __builtin_amdgcn_sched_barrier(0);  // nothing that waits for an earlier load may move above the ring's requests
  };
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };  // LDS-only: must not drain the ring
  if constexpr (FUSE == 1) {
    // M == 1: this wave's k-run is blocks [kb0, kb1): (kb1 - kb0) x 16 vectors of 8 bf16, at most XW per lane (host: <= 16 blocks per wave)
    constexpr int XW = 4;
    const int nv = (kb1 - kb0) * 16;
    const u32x4* xr = reinterpret_cast<const u32x4*>(xq) + (size_t)kb0 * 16;  // (xq: the bf16 activation)
    u32x4 xv[XW];
#pragma unroll
    for (int i = 0; i < XW; ++i) xv[i] = xr[max(min(lane + i * 64, nv - 1), 0)];
    __builtin_amdgcn_sched_barrier(0);
    prime();
    float m = 0.f;
    bool has_nan = false;
#pragma unroll
    for (int i = 0; i < XW; ++i) m = fmaxf(m, amax8(xv[i], has_nan));
    if (has_nan) m = INFINITY;
    m = wave_max(m);
    wmax[wave * 16] = m;
    lds_barrier();
    float mm = 0.f;
    for (int w = 0; w < nwaves; ++w) mm = fmaxf(mm, wmax[w * 16]);
    const float sc = fp8_row_scale(mm);
    rs[0] = sc;  // (every thread: same value)
#pragma unroll
    for (int i = 0; i < XW; ++i) {
      const int idx = lane + i * 64;
      if (idx < nv) *reinterpret_cast<u32x2*>(codes + kb0 * 128 + idx * 8) = fp8_quant8(xv[i], sc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private codes: no barrier
  } else if constexpr (FUSE == 2) {
    const int nvec = K >> 3, nthreads = blockDim.x, tid = threadIdx.x;
    const uint16_t* xb = reinterpret_cast<const uint16_t*>(xq) + (size_t)m0 * K;
    for (int r = 0; r < rows; ++r) {
      const u32x4* xr = reinterpret_cast<const u32x4*>(xb + (size_t)r * K);
      float m = 0.f;
      bool has_nan = false;
      for (int i = tid; i < nvec; i += nthreads) m = fmaxf(m, amax8(xr[i], has_nan));
      if (has_nan) m = INFINITY;
      m = wave_max(m);
      wmax[wave * 16 + r] = m;
    }
    lds_barrier();
    if (tid < rows) {
      float m = 0.f;
      for (int w = 0; w < nwaves; ++w) m = fmaxf(m, wmax[w * 16 + tid]);
      rs[tid] = fp8_row_scale(m);
    }
    lds_barrier();
    for (int r = 0; r < rows; ++r) {
      const u32x4* xr = reinterpret_cast<const u32x4*>(xb + (size_t)r * K);
      const float sc = rs[r];
      for (int i = tid; i < nvec; i += nthreads) *reinterpret_cast<u32x2*>(codes + r * cstride + i * 8) = fp8_quant8(xr[i], sc);
    }
    prime();
    lds_barrier();
  } else {
    prime();
  }
  auto for_slots = [&](auto&& f) {
    [&]<int... D>(std::integer_sequence<int, D...>) { (f(std::integral_constant<int, D>{}), ...); }(std::make_integer_sequence<int, DEPTH>{});
  };
  int kb = kb0;
  for (; kb + 2 * DEPTH <= kb1; kb += DEPTH) {
    for_slots([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      consume(st[d], kb + d);
      issue(st[d], kb + d + DEPTH);
    });
  }
  for_slots([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    if (kb + d < kb1) {
      consume(st[d], kb + d);
      if (kb + d + DEPTH < kb1) issue(st[d], kb + d + DEPTH);
    }
  });
  kb += DEPTH;
  for_slots([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    if (kb + d < kb1) consume(st[d], kb + d);
  });

  // cross-wave reduction: red[wave][m-tile][n-tile][row][col]; D layout: lane (col = lane & 15, group kq) holds rows 4 kq + {0..3}
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float* r = red + ((wave * MT + mt) * NT + nt) * 256 + (kq * 4) * 16 + (lane & 15);
      r[0] = acc[mt][nt].x; r[16] = acc[mt][nt].y; r[32] = acc[mt][nt].z; r[48] = acc[mt][nt].w;
    }
  __syncthreads();
  const int tid = threadIdx.x;
  for (int e = tid; e < MT * NT * 256; e += blockDim.x) {
    const int tile = e >> 8, mt = tile / NT, nt = tile - mt * NT;
    const int row = 16 * mt + ((e >> 4) & 15), col = 16 * nt + (e & 15);
    if (row < rows) {
      float sum = 0.f;
      for (int w = 0; w < nwaves; ++w) sum += red[w * (MT * NT * 256) + e];
      float v = sum * (FUSE != 0 ? rs[row] : x_scale[m0 + row]);
      if (bias != nullptr) v += bf16_lo_to_f32(bias[ntile * 16 + col]);
      y[(size_t)(m0 + row) * N + ntile * 16 + col] = f32_to_bf16_bits(v);
    }
  }
}

// product rule of the unfused form: m-tiles and n-tiles per workgroup (see launch_fp8_int4)
int fp8_int4_m_tiles(int64_t M) { return M > 16 ? 2 : 1; }
int fp8_int4_n_tiles(int64_t M, int64_t N, int64_t K, int group_size) {
  // (groups of 32 / 64: two n-tiles carry 4 / 2 (scale, zero) words per n-tile and ring stage and spill 12 VGPRs -- one n-tile there)
  return (N % 32 == 0 && group_size >= 128 && (M > 64 || (M > 32 && K >= 8192))) ? 2 : 1;
}

template <int G>
int launch_fp8_int4(const uint8_t* xq, const float* x_scale, const int32_t* qdata, const uint16_t* sz, const uint16_t* bias, uint16_t* y,
                    int64_t M, int64_t N, int64_t K, hipStream_t stream, bool fused) {
  const int kblocks = (int)(K >> 7);
  int wpb = (kblocks >= 16) ? 8 : 4;
  if (wpb > kblocks) wpb = kblocks < 4 ? 4 : kblocks;
  size_t smem = (size_t)wpb * (16 * (128 + 16) + 1024);
  dim3 grid((unsigned)(N >> 4), (unsigned)((M + 15) / 16)), block(wpb * 64);
  const u32x4* qd = reinterpret_cast<const u32x4*>(qdata);
  const uint32_t* szw = reinterpret_cast<const uint32_t*>(sz);
  if (!fused) {
    // Tiles per workgroup (profiles/fp8_int4_mt_ab_r05.jsonl, fp8_int4_mt_nt_ab_r05.jsonl; cold weights, us; every form gives the same bits):
    // two m-tiles from 17 rows (gate_proj at M = 128: 64.6 -> 46.3), and two n-tiles as well above 64 rows, or above 32 on K >= 8192
    // (down_proj 4096 x 14336 at M = 128 / 256 / 512: 47.4 -> 33.8, 90.4 -> 67.0, 177 -> 132; qkv at M = 512 76.4 -> 73.5; at M <= 64 on
    // K = 4096 the halved grid costs 1 us).  Four m-tiles x one n-tile (2-deep ring: the x stages of a deeper one cost the occupancy;
    // 3-deep measured 5 - 12 % slower) lose to 2 x 2 in every cell from 64 rows and stay a tuning form; 4 x 2 spills.
    // g_fp8_int4_mt / g_fp8_int4_nt1 (ao_int4_set_tuning modes 961 / 962 / 964, 972 / 974) force a form for A/B runs.
    const bool forced = g_fp8_int4_mt == 1 || g_fp8_int4_mt == 2 || g_fp8_int4_mt == 4;
    const int mt = forced ? g_fp8_int4_mt : fp8_int4_m_tiles(M);
    const bool nt2 = forced ? ((N % 32 == 0) && !g_fp8_int4_nt1) : (!g_fp8_int4_nt1 && fp8_int4_n_tiles(M, N, K, G) == 2);
    const size_t slab = (size_t)wpb * 16 * (128 + 16), redb = (size_t)wpb * 1024;
    auto go = [&](auto kern, int mtv, int ntv) -> int {
      const size_t sm = slab * mtv + redb * mtv * ntv;
      dim3 g((unsigned)(N / (16 * ntv)), (unsigned)((M + 16 * mtv - 1) / (16 * mtv)));
      if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), sm, "hipFuncSetAttribute(fp8_int4_mm_kernel)")) return rc;
      ao::launch(kern, g, block, sm, stream, xq, x_scale, qd, szw, bias, y, (int)M, (int)N, (int)K);
      AO_LAUNCH_CHECK("fp8_int4_mm_kernel launch");
      return AO_OK;
    };
    if (mt == 4) return go(fp8_int4_mm_kernel<G, 2, 0, 4, 1>, 4, 1);  // (four m-tiles x two n-tiles do not fit 256 VGPRs without spilling)
    if (mt == 2) return nt2 ? go(fp8_int4_mm_kernel<G, 3, 0, 2, 2>, 2, 2) : go(fp8_int4_mm_kernel<G, 3, 0, 2, 1>, 2, 1);
    ao::launch(fp8_int4_mm_kernel<G, 4, 0>, grid, block, smem, stream, xq, x_scale, qd, szw, bias, y, (int)M, (int)N, (int)K);
  } else {
    smem += (size_t)((M * (K + 16) + 15) & ~(int64_t)15) + (size_t)(wpb * 16 + 16) * sizeof(float);
    // M == 1 with at most 16 blocks per wave: the wave-private form; else the workgroup-wide cast
    const bool priv = M == 1 && (kblocks + wpb - 1) / wpb <= 16;
    const void* kern = priv ? reinterpret_cast<const void*>(fp8_int4_mm_kernel<G, 4, 1>) : reinterpret_cast<const void*>(fp8_int4_mm_kernel<G, 4, 2>);
    if (int rc = ensure_dynamic_lds(kern, smem, "hipFuncSetAttribute(fp8_int4_mm_kernel)")) return rc;
    if (priv) ao::launch(fp8_int4_mm_kernel<G, 4, 1>, grid, block, smem, stream, xq, x_scale, qd, szw, bias, y, (int)M, (int)N, (int)K);
    else ao::launch(fp8_int4_mm_kernel<G, 4, 2>, grid, block, smem, stream, xq, x_scale, qd, szw, bias, y, (int)M, (int)N, (int)K);
  }
  AO_LAUNCH_CHECK("fp8_int4_mm_kernel launch");
  return AO_OK;
}

int fp8_int4_dispatch(const uint8_t* xq, const float* x_scale, const int32_t* qdata, const uint16_t* sz, const uint16_t* bias, uint16_t* y, int64_t M,
                      int64_t N, int64_t K, int group_size, hipStream_t s, bool fused) {
  switch (group_size) {
    case 32: return launch_fp8_int4<32>(xq, x_scale, qdata, sz, bias, y, M, N, K, s, fused);
    case 64: return launch_fp8_int4<64>(xq, x_scale, qdata, sz, bias, y, M, N, K, s, fused);
    case 128: return launch_fp8_int4<128>(xq, x_scale, qdata, sz, bias, y, M, N, K, s, fused);
    default: return launch_fp8_int4<256>(xq, x_scale, qdata, sz, bias, y, M, N, K, s, fused);
  }
}

int fp8_int4_check(const char* fn, int64_t M, int64_t N, int64_t K, int group_size) {
  AO_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0, "%s: N=%lld must be a multiple of 16 and K=%lld of 128", fn, (long long)N, (long long)K);
  AO_REQUIRE(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256, "%s: group_size must be 32, 64, 128 or 256, got %d", fn,
             group_size);
  AO_REQUIRE(K % group_size == 0, "%s: K=%lld not divisible by group_size=%d", fn, (long long)K, group_size);
  AO_REQUIRE(M >= 0 && M < (1ll << 20) && N < (1ll << 31) && K < (1ll << 31), "%s: bad M=%lld", fn, (long long)M);
  return AO_OK;
}

}  // namespace
void fp8_int4_set_mt(int mt, bool nt1) { g_fp8_int4_mt = mt; g_fp8_int4_nt1 = nt1; }
}  // namespace ao

using namespace ao;

extern "C" int ao_fp8_int4_linear(const uint8_t* xq, const float* x_scale, const int32_t* qdata, const uint16_t* scale_and_zero,
                                  const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K, int group_size, void* stream) {
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE_PTR(scale_and_zero);
  if (int rc = fp8_int4_check(__func__, M, N, K, group_size)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(xq);
  AO_REQUIRE_PTR(x_scale);
  AO_REQUIRE_PTR(y);
  return fp8_int4_dispatch(xq, x_scale, qdata, scale_and_zero, bias, y, M, N, K, group_size, (hipStream_t)stream, false);
}

extern "C" const char* ao_fp8_int4_kernel_name(int64_t M, int64_t N, int64_t K, int group_size) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 16 != 0 || K % 128 != 0 || !(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256) ||
      K % group_size != 0)
    return "invalid";
  if (fp8_int4_m_tiles(M) == 1) return "fp8_int4_mm_kernel<1x1>";
  return fp8_int4_n_tiles(M, N, K, group_size) == 2 ? "fp8_int4_mm_kernel<2x2>" : "fp8_int4_mm_kernel<2x1>";
}

extern "C" int ao_fp8_int4_dynamic_fits(int64_t M, int64_t N, int64_t K) {
  return (M >= 1 && M <= 16 && N % 16 == 0 && K % 128 == 0 && M * (K + 16) <= 64 * 1024) ? 1 : 0;
}

extern "C" int ao_fp8_int4_dynamic_linear(const uint16_t* x, const int32_t* qdata, const uint16_t* scale_and_zero, const uint16_t* bias, uint16_t* y,
                                          int64_t M, int64_t N, int64_t K, int group_size, void* stream) {
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE_PTR(scale_and_zero);
  if (int rc = fp8_int4_check(__func__, M, N, K, group_size)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE(ao_fp8_int4_dynamic_fits(M, N, K), "%s: the fused form holds the cast activation in LDS: M <= 16 and M * (K + 16) <= 65536, got M=%lld K=%lld "
             "(use ao_fp8_quantize_rowwise + ao_fp8_int4_linear)", __func__, (long long)M, (long long)K);
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(y);
  return fp8_int4_dispatch(reinterpret_cast<const uint8_t*>(x), nullptr, qdata, scale_and_zero, bias, y, M, N, K, group_size, (hipStream_t)stream, true);
}
