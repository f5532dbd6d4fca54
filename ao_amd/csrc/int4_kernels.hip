// int4 weight-only ("tinygemm", tile-packed-to-4d) kernels for gfx950.
//
// Weight format (bit-exact with aten::_convert_weight_to_int4pack on ROCm, the
// on-disk format of Int4TilePackedTo4dTensor): qdata is int32
// [N/8][K/128][32][4] by shape, but indexed on ROCm as [N/16][K/128][64][4]:
// one 1 KiB block = one wavefront-load (64 lanes x 16 B) = a 16(n) x 128(k)
// tile.  Lane l owns n = l & 15 and, in each of the 8 inner k-tiles (16 k
// each), the 4 consecutive k's starting at (l >> 4) * 4.  Word j of the lane
// packs inner k-tiles 2j and 2j+1 as v0|v2<<4|v4<<8|v6<<12|v1<<16|v3<<20|v5<<24|v7<<28.
// That ownership is exactly the B-operand ownership of the 16x16 MFMA family, so
// a dequantised word is a ready v_mfma_f32_16x16x32_bf16 B fragment
// (k-slots 0-3 = tile 2j, 4-7 = tile 2j+1; the A fragment uses the same slots).
//
// Reference call sites replaced (torchao 0.19.0 snapshot):
//   int4_tile_packed_to_4d_tensor.py:202  aten::_convert_weight_to_int4pack
//   int4_tile_packed_to_4d_tensor.py:287  aten::_weight_int4pack_mm
//   quant_primitives.py:999-1007          dequant rounding sequence (the oracle)
//   quant_primitives.py:1299-1335,577-599 tinygemm qparams / quantize
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"
#include "lds_dma.h"
#include "splitk.h"

namespace ao {
void fp8_rowwise_rb_set_trace(unsigned long long* p);  // fp8_rb_kernels.hip
namespace {

// ---------------------------------------------------------------------------
// Dequantise one packed word (8 nibbles) into four packed-bf16 pairs
// (w0w1, w2w3 | w4w5, w6w7) with the oracle's rounding sequence:
//     w = bf16( bf16( (q - 8) * s ) + z )
// (q-8)*s is exact in fp32 (4-bit x 8-bit significands), so the first
// v_cvt_pk_bf16_f32 is the single rounding torch's bf16 multiply performs; the
// fp32 add + second v_cvt_pk_bf16_f32 is torch's bf16 add.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void dequant_word(uint32_t p, float s, float neg8s, float z,
                                             uint32_t (&out)[4]) {
  uint32_t lo = p & 0x0F0F0F0Fu;         // bytes: v0, v4, v1, v5
  uint32_t hi = (p >> 4) & 0x0F0F0F0Fu;  // bytes: v2, v6, v3, v7
  // keep the masked words opaque so the byte extracts lower to v_cvt_f32_ubyteN
  asm("" : "+v"(lo));
  asm("" : "+v"(hi));
  const float f0 = (float)(lo & 0xffu), f4 = (float)((lo >> 8) & 0xffu);
  const float f1 = (float)((lo >> 16) & 0xffu), f5 = (float)(lo >> 24);
  const float f2 = (float)(hi & 0xffu), f6 = (float)((hi >> 8) & 0xffu);
  const float f3 = (float)((hi >> 16) & 0xffu), f7 = (float)(hi >> 24);
  const f32x2 q[4] = {{f0, f1}, {f2, f3}, {f4, f5}, {f6, f7}};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 t = q[i] * s + neg8s;                 // exact (q-8)*s
    const uint32_t tp = pack_bf16x2(t.x, t.y);        // rounding #1
    const f32x2 tr = {bf16_lo_to_f32(tp), bf16_hi_to_f32(tp)};
    const f32x2 w = tr + z;                           // fp32 add, like torch
    out[i] = pack_bf16x2(w.x, w.y);                   // rounding #2
  }
}

// ---------------------------------------------------------------------------
// The product form of the same exact sequence (the all-VALU form above serves ao_int4_dequantize).  All VALU instructions cost ~4.4 issue cycles per wave64 on gfx950
// (profiles/ubench_valu_r01.txt) and the matrix pipe is idle in a GEMV, so:
//   * q -> fp32: a byte holding q (0..15) read as OCP e4m3 is q * 2^-9, so
//     v_cvt_scalef32_pk_f32_fp8 with scale 2^9 converts two nibbles per instruction, exactly;
//   * the IEEE add t + z runs on the matrix pipe:  D = I * B + C  with an identity A operand
//     returns each lane's own four bf16 values widened to fp32 plus C (one product per output, so
//     the fp32 result is the correctly rounded sum).
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 identity_fragment(int lane) {
  // row (lane & 3) of the 4x4 identity as a 4x4x4 A operand (4 bf16, k = 0..3)
  const int hot = lane & 3;
  s16x4 f;
  f.x = hot == 0 ? (short)0x3F80 : (short)0;
  f.y = hot == 1 ? (short)0x3F80 : (short)0;
  f.z = hot == 2 ? (short)0x3F80 : (short)0;
  f.w = hot == 3 ? (short)0x3F80 : (short)0;
  return f;
}

__device__ __forceinline__ f32x4 widen_add(s16x4 ident, uint32_t lo_pair, uint32_t hi_pair, f32x4 c) {
  const u32x2 bb = {lo_pair, hi_pair};
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ident, __builtin_bit_cast(s16x4, bb), c, 0, 0, 0);
}

__device__ __forceinline__ void dequant_word_mfma(uint32_t p, float s, float neg8s, float z, s16x4 ident,
                                                  uint32_t (&out)[4]) {
  const uint32_t lo = p & 0x0F0F0F0Fu;         // bytes: v0, v4, v1, v5
  const uint32_t hi = (p >> 4) & 0x0F0F0F0Fu;  // bytes: v2, v6, v3, v7
  const f32x2 r0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, false);  // q0, q4
  const f32x2 r1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, true);   // q1, q5
  const f32x2 r2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, false);  // q2, q6
  const f32x2 r3 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, true);   // q3, q7
  const f32x2 t0 = r0 * s + neg8s, t1 = r1 * s + neg8s, t2 = r2 * s + neg8s, t3 = r3 * s + neg8s;
  const uint32_t tp0 = pack_bf16x2(t0.x, t1.x), tp1 = pack_bf16x2(t2.x, t3.x);  // t = bf16((q-8)*s): rounding #1
  const uint32_t tp2 = pack_bf16x2(t0.y, t1.y), tp3 = pack_bf16x2(t2.y, t3.y);
  const f32x4 zz = {z, z, z, z};
  const f32x4 w0 = widen_add(ident, tp0, tp1, zz);  // fl32(t + z), lanes' own values
  const f32x4 w1 = widen_add(ident, tp2, tp3, zz);
  out[0] = pack_bf16x2(w0.x, w0.y); out[1] = pack_bf16x2(w0.z, w0.w);  // rounding #2
  out[2] = pack_bf16x2(w1.x, w1.y); out[3] = pack_bf16x2(w1.z, w1.w);
}

// dequant_word_mfma cut into four stages so that a caller can slot them between independent MFMAs
// (the batched register-B kernel: one wave per SIMD, nothing else hides the VALU work)
struct DequantPipe {
  f32x2 r0, r1, r2, r3;
  uint32_t tp0, tp1, tp2, tp3;
  f32x4 w0, w1;
  uint32_t out[4];
};
template <int STAGE>
__device__ __forceinline__ void dequant_stage(DequantPipe& d, uint32_t p, float s, float neg8s, float z, s16x4 ident) {
  if constexpr (STAGE == 0) {
    const uint32_t lo = p & 0x0F0F0F0Fu, hi = (p >> 4) & 0x0F0F0F0Fu;
    d.r0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, false);
    d.r1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, true);
    d.r2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, false);
    d.r3 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, true);
  } else if constexpr (STAGE == 1) {
    const f32x2 t0 = d.r0 * s + neg8s, t1 = d.r1 * s + neg8s, t2 = d.r2 * s + neg8s, t3 = d.r3 * s + neg8s;
    d.tp0 = pack_bf16x2(t0.x, t1.x); d.tp1 = pack_bf16x2(t2.x, t3.x);  // rounding #1
    d.tp2 = pack_bf16x2(t0.y, t1.y); d.tp3 = pack_bf16x2(t2.y, t3.y);
  } else if constexpr (STAGE == 2) {
#ifdef AO_DEQUANT_VALU  // profiling build (tools/bin/_C_mi355_valu.so): the t + z add on the VALU instead of the matrix pipe -- 12 VALU for 2 MFMAs per word
    d.w0 = f32x4{bf16_lo_to_f32(d.tp0) + z, bf16_hi_to_f32(d.tp0) + z, bf16_lo_to_f32(d.tp1) + z, bf16_hi_to_f32(d.tp1) + z};
    d.w1 = f32x4{bf16_lo_to_f32(d.tp2) + z, bf16_hi_to_f32(d.tp2) + z, bf16_lo_to_f32(d.tp3) + z, bf16_hi_to_f32(d.tp3) + z};
#else
    const f32x4 zz = {z, z, z, z};
    d.w0 = widen_add(ident, d.tp0, d.tp1, zz);
    d.w1 = widen_add(ident, d.tp2, d.tp3, zz);
#endif
  } else {
    d.out[0] = pack_bf16x2(d.w0.x, d.w0.y); d.out[1] = pack_bf16x2(d.w0.z, d.w0.w);  // rounding #2
    d.out[2] = pack_bf16x2(d.w1.x, d.w1.y); d.out[3] = pack_bf16x2(d.w1.z, d.w1.w);
  }
}

// ---------------------------------------------------------------------------
// y[M,N] = x[M,K] @ dequant(qdata)^T     (aten::_weight_int4pack_mm)
//
// grid = (N/16, ceil(M/16)); block = WPB waves.  A workgroup owns one 16-wide
// n-tile for one slab of <=16 rows; its waves split K into contiguous ranges of
// 1 KiB weight blocks.  Each wave streams its blocks straight into VGPRs
// (non-temporal dwordx4, DEPTH blocks in flight), dequantises in registers and
// feeds v_mfma_f32_16x16x32_bf16.  x never goes through a block-wide barrier:
// each wave stages the 128-k slice it needs into a private LDS slab in fragment
// order and reads it back as A fragments (rows >= M alias one zero row).
// One barrier at the end for the cross-wave (split-K) reduction.
// ---------------------------------------------------------------------------
template <int MAXM>
struct XRegs {
  // MAXM <= 4: one dword (2 bf16) per row per lane;  MAXM > 4: MAXM/4 x 16 B
  static constexpr int kDwords = (MAXM <= 4) ? MAXM : MAXM;  // MAXM/4 * 4 dwords
  uint32_t v[kDwords];
};

// A block is consumed word by word (dequantise one packed word, multiply, next word).  Rounds 2 - 3 also built, measured and dropped
// (profiles/int4_modes_r03.jsonl, dequant_lab_r02.txt; removed from the source in round 6): stage-by-stage and software-pipelined
// instruction orders of the dequant, several n-tiles per workgroup, and cross-workgroup K parts at M = 1.
// STRAIGHT: every wave of the workgroup owns exactly DEPTH blocks (K = 128 DEPTH waves): prologue, one pass of the ring, done --
//   no steady-state / drain loops, fewer live registers (64 VGPRs: four 8-wave workgroups per CU instead of three)
template <int G, int MAXM, int DEPTH, bool STRAIGHT = false>
__global__ __launch_bounds__((MAXM > 4) ? 512 : 1024) void int4_mm_kernel(
    const uint16_t* __restrict__ x, const u32x4* __restrict__ qdata,
    const uint32_t* __restrict__ sz, uint16_t* __restrict__ y, int M, int N, int K, float* __restrict__ /* unused */, unsigned* __restrict__ /* unused */) {
  // (the two trailing arguments carried the M = 1 split-K workspace of round 3; the form is gone, the kernarg layout stays: removing them
  // re-allocates the prologue's scalar registers, and every measurement of this kernel since round 3 was taken on this layout)
  constexpr int NG = (G >= 128) ? 1 : (128 / G);              // groups per 128-k block
  constexpr int ROWSTRIDE = (MAXM <= 4) ? 256 : 272;          // bytes, padded vs bank conflicts
  constexpr int SLAB = (MAXM + 1) * ROWSTRIDE;                // per-wave x staging: rows 0 .. MAXM - 1 + one zero row
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int ntile = blockIdx.x;
  const int m0 = blockIdx.y * 16;
  const int rows = min(16, M - m0);
  const int kblocks = K >> 7;
  // straight-line form: every wave owns exactly DEPTH blocks (the host checked K = 128 DEPTH waves) -- no division by the runtime wave count
  constexpr bool kFixedRun = STRAIGHT;
  const int kb0 = kFixedRun ? wave * DEPTH : (kblocks * wave) / nwaves;
  const int kb1 = kFixedRun ? kb0 + DEPTH : (kblocks * (wave + 1)) / nwaves;

  char* slab = smem + wave * SLAB;
  float* red = reinterpret_cast<float*>(smem + nwaves * SLAB);

  // zero row (last row) of this wave's slab: 256 B
  *reinterpret_cast<uint32_t*>(slab + MAXM * ROWSTRIDE + lane * 4) = 0u;

  const int n = ntile * 16 + (lane & 15);
  const int kq = lane >> 4;
  const u32x4* wp = qdata + (size_t)ntile * kblocks * 64 + lane;
  // the same stream as a wave-uniform base + a 32-bit lane offset (global_load saddr form: no 64-bit VALU address per request)
  // straight-line form: the weight stream through a buffer descriptor -- SGPR base + SGPR block offset + one 32-bit lane offset, so a
  // request costs no 64-bit VALU address
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(qdata) + (size_t)ntile * kblocks * 1024), 0, kblocks * 1024, 0x00020000);
  const uint32_t wlane = (uint32_t)lane * 16u;
  const uint16_t* xrow0 = x + (size_t)m0 * K;

  // LDS addresses
  const int mrow = lane & 15;
  const bool a_zero = !(mrow < rows);
  const char* a_base = slab + (a_zero ? MAXM : mrow) * ROWSTRIDE + kq * 64;
  int st_off;  // byte offset (within a row) this lane stores its x piece at
  if (MAXM <= 4) {
    // dword = k = 2*lane, 2*lane+1
    st_off = ((lane >> 1) & 3) * 64 + (lane >> 4) * 16 + (((lane >> 3) & 1) * 4 + (lane & 1) * 2) * 2;
  } else {
    // 16-B piece c = lane & 15 of row (lane >> 4) + 4*i: k = 8c .. 8c+7
    const int c = lane & 15;
    st_off = ((c & 1) * 2) * 64 + (c >> 2) * 16 + (((c >> 1) & 1) * 4) * 2;
  }

  struct Stage {
    u32x4 w;
    uint32_t sz[NG];
    XRegs<MAXM> xr;
  };
  Stage st[DEPTH];

  // Loads are unconditional (rows beyond `rows` re-read the last valid row; the
  // LDS copy of such a row is never read) so that the steady-state loop is
  // straight-line code and the compiler can use counted s_waitcnt vmcnt(N).
  const int last_row = rows - 1;
  auto issue = [&](Stage& s, int kb) {
    const int kg0 = (G >= 128) ? ((kb * 128) / G) : (kb * NG);
    if constexpr (kFixedRun && MAXM == 1) {
      // measured (profiles/int4_addr_ab_r04.txt, three alternating processes on one box): weights through the descriptor 812 -> 818 tok/s;
      // the scale / zero word and the x slice through descriptors too 796 -- those two stay plain global loads
      s.w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)wlane, kb * 1024, 2 /* nt: streamed once */));
#pragma unroll
      for (int i = 0; i < NG; ++i) s.sz[i] = sz[(size_t)(kg0 + i) * N + n];
      s.xr.v[0] = *reinterpret_cast<const uint32_t*>(xrow0 + (size_t)kb * 128 + lane * 2);
      return;
    }
    s.w = __builtin_nontemporal_load(wp + (size_t)kb * 64);
#pragma unroll
    for (int i = 0; i < NG; ++i) s.sz[i] = sz[(size_t)(kg0 + i) * N + n];
    if (MAXM <= 4) {
#pragma unroll
      for (int r = 0; r < MAXM; ++r) {
        const int rr = min(r, last_row);
        s.xr.v[r] = *reinterpret_cast<const uint32_t*>(xrow0 + (size_t)rr * K + kb * 128 + lane * 2);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MAXM / 4; ++i) {
        const int rr = min((lane >> 4) + 4 * i, last_row);
        const u32x4 t = *reinterpret_cast<const u32x4*>(xrow0 + (size_t)rr * K + kb * 128 + (lane & 15) * 8);
        s.xr.v[4 * i + 0] = t.x; s.xr.v[4 * i + 1] = t.y;
        s.xr.v[4 * i + 2] = t.z; s.xr.v[4 * i + 3] = t.w;
      }
    }
  };

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const s16x4 ident = identity_fragment(lane);

  auto stage_x = [&](const Stage& s, char* dst) {
    // stage x (wave-private: DS ops of one wave execute in order, no barrier)
    if (MAXM <= 4) {
#pragma unroll
      for (int r = 0; r < MAXM; ++r)
        *reinterpret_cast<uint32_t*>(dst + r * ROWSTRIDE + st_off) = s.xr.v[r];
    } else {
#pragma unroll
      for (int i = 0; i < MAXM / 4; ++i) {
        const int r = (lane >> 4) + 4 * i;
        char* d = dst + r * ROWSTRIDE + st_off;
        *reinterpret_cast<u32x2*>(d) = u32x2{s.xr.v[4 * i + 0], s.xr.v[4 * i + 1]};
        *reinterpret_cast<u32x2*>(d + 64) = u32x2{s.xr.v[4 * i + 2], s.xr.v[4 * i + 3]};
      }
    }
  };
  auto mma = [&](const u32x4& a, const uint32_t (&b)[4], f32x4& c) {
    const u32x4 bv = {b[0], b[1], b[2], b[3]};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
  };

  auto consume = [&](const Stage& s) {
    const uint32_t wds[4] = {s.w.x, s.w.y, s.w.z, s.w.w};
    stage_x(s, slab);
    u32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const u32x4*>(a_base + j * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gi = (G >= 128) ? 0 : ((j * 32) / G);
      const float sc = bf16_lo_to_f32(s.sz[gi]);
      const float zp = bf16_hi_to_f32(s.sz[gi]);
      uint32_t b[4];
      dequant_word_mfma(wds[j], sc, -8.0f * sc, zp, ident, b);
      mma(a[j], b, acc);
    }
  };

  // prologue: DEPTH blocks in flight (indices clamped into the wave's range; a
  // wave with fewer than DEPTH blocks just re-reads its last one, unused)
  const int kb_last = max(kb1 - 1, kb0);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(st[d], kFixedRun ? kb0 + d : min(kb0 + d, kb_last));

  // ring slots as compile-time indices
  auto for_slots = [&](auto&& f) {
    [&]<int... D>(std::integer_sequence<int, D...>) { (f(std::integral_constant<int, D>{}), ...); }(std::make_integer_sequence<int, DEPTH>{});
  };
  if constexpr (STRAIGHT) {
    // the host guarantees kb1 - kb0 == DEPTH for every wave: one pass of the ring
    for_slots([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      consume(st[d]);
    });
    float* r = red + wave * 256 + (kq * 4) * 16 + (lane & 15);
    r[0] = acc.x; r[16] = acc.y; r[32] = acc.z; r[48] = acc.w;
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
  int kb = kb0;
  // steady state: every consumed stage is refilled, no branches in the body
  for (; kb + 2 * DEPTH <= kb1; kb += DEPTH) {
    for_slots([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      consume(st[d]);
      issue(st[d], kb + d + DEPTH);
    });
  }
  // drain: fewer than 2*DEPTH blocks left
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
  }

  // cross-wave reduction: red[tile][wave][row][col]
  if constexpr (!STRAIGHT) {
    float* r = red + wave * 256 + (kq * 4) * 16 + (lane & 15);
    r[0] = acc.x; r[16] = acc.y; r[32] = acc.z; r[48] = acc.w;
  }
  __syncthreads();
  const int tid = threadIdx.x;
  const int row = tid >> 4, col = tid & 15;
  if (tid < 256 && row < rows) {
    float sum = 0.f;
    for (int w = 0; w < nwaves; ++w) sum += red[w * 256 + tid];
    y[(size_t)(m0 + row) * N + ntile * 16 + col] = f32_to_bf16_bits(sum);
  }
}

// ---------------------------------------------------------------------------
// 16 < M (batched decode / small prefill, the "bs = 128" half of the BASELINE metric), "register-B" form: each packed
// block is dequantised once per 128-row slab and the dequantised weights never touch LDS -- a wave that owns NT n-tiles
// dequantises their packed words straight into B operands of v_mfma_f32_16x16x32_bf16 and multiplies all 128 rows
// of the slab against them; only x goes through LDS, once per workgroup and k-block.
//
// Operand shapes.  In the packed block lane (n, kq) owns, per word j, the runs k = 32j + 16h + 4kq + {0..3}
// (h = 0, 1): never 8 contiguous k, so an A operand matching "one lane = one packed lane" would be two 8-byte
// pieces of x (ds_read2_b64: half the LDS rate, and LDS-DMA cannot scatter 8-byte pieces).  The weights sit in
// LDS anyway (ring below), so a lane reads a different 16 bytes of the block instead: lane (n, g), g = 2a + t,
// takes words {2a, 2a + 1} of packed lanes (n, kq = 2t) and (n, kq = 2t + 1).  For e, h in {0, 1} its runs
// (j = 2a + e, h) of the two kq are adjacent: k = 64a + 32e + 16h + 8t + {0..7}.  The MFMA of phase p = 2e + h
// sums the 4 lane groups' 8 k = all (a, t): 32 distinct k; four phases cover the 128.  The matching A operand is
// one aligned 16-byte chunk of x: chunk c = 8a + 4e + 2h + t = 2p ^ (8a | t)  ->  one ds_read_b128.
//   * workgroup = WAVES waves, tile 128 (m) x WAVES NT 16 (n), grid.z = split-K parts;
//   * x tile [128 rows][256 B] by LDS-DMA (16 B per lane, 4 rows per instruction).  DMA writes lane-linear, so rows
//     cannot be padded; the SOURCE is swizzled instead: LDS chunk position c' of row r holds global chunk c' ^ (r & 15),
//     which makes every 16-lane service group of the b128 fragment reads hit 16 different bank windows;
//   * packed weights + scale/zero words by LDS-DMA into a wave-private ring (as in the M = 1 kernel);
//   * one ring of 3 stages for everything, filled two k-blocks ahead, one s_waitcnt vmcnt(LPS) + one LDS-only
//     barrier per k-block (all loads are DMAs issued from asm: the compiler's own counting never sees them);
//   * the k-block is scheduled by hand in slots of two MFMAs: next A operands are read, the next two words are
//     dequantised (four VALU stages per word) and the ring's DMAs are dealt one per slot (issued back to back they
//     queue in the texture path and hold the wave ~120 cycles each; spreading them further or staggering them
//     between waves measured 7 % slower).
// ---------------------------------------------------------------------------
constexpr int kRbStages = 3;  // x ring (shared by the workgroup): filled two k-blocks ahead
// weight ring (per wave), filled kW - 1 k-blocks ahead.  Its depth is independent of the x ring's; deepening it to 6 - 8 stages
// (what fixed the 8-bit weight-streaming kernel, rb8_kernels.hip) measured 0.96x here (bs = 128: 31.9k -> 30.8k tok/s,
// profiles/int4_bs_sweep_r02.txt): at one wave per SIMD the k-block is bound by what the wave itself issues (10 LDS-DMAs, 32
// ds_read_b128, ~70 VALU, 32 MFMAs, one after the other), not by the HBM stream's latency.  So: 3 stages.
constexpr int kRbWeightStages = 3;
constexpr int rb_wstages(int waves, int mt, int wst) {
  const int room = (160 * 1024 - kRbStages * mt * 4096) / (waves * wst);
  return room >= kRbWeightStages ? kRbWeightStages : (room < 3 ? 3 : room);
}

// ABL (profiling builds only): 1 no 16x16x32 MFMAs, 2 no dequant, 3 no A reads, 4 no DMAs, 6 no scale / zero DMAs, 5 product + s_memtime stamps of wave 0
// (16 u64 per workgroup: entry, ring primed, barrier of k-blocks 0..7 passed, loop done, meeting done, exit)
// MT = 16-row m-tiles per slab (8, 4, 2, 1): batches below 128 rows stage, read and multiply only the rows they have.
// PROD: WAVES more waves that do nothing but issue the ring's DMAs (producer wave WAVES + w feeds consumer wave w: its weight ring and
// its quarter of the x tile).  A global_load_lds issue holds its wave for 50 - 120 cycles, ten of them per k-block; with one or two
// in-order waves per SIMD nothing else of that wave issues meanwhile -- on their own waves they cost the consumers nothing.
template <int G, int WAVES, int NT, int MT = 8, int ABL = 0, bool PROD = false>
__global__ __launch_bounds__(64 * WAVES * (PROD ? 2 : 1), (WAVES == 4 && !PROD) ? 2 : 1) void int4_mm_rb_kernel(
    const uint16_t* __restrict__ x, const u32x4* __restrict__ qdata, const uint32_t* __restrict__ sz, uint16_t* __restrict__ y, int M,
    int N, int K, float* __restrict__ ws, unsigned* __restrict__ tickets, unsigned long long* __restrict__ trace) {
  unsigned long long ts[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (ABL == 5) ts[0] = __builtin_amdgcn_s_memtime();
  constexpr int NG = (G >= 128) ? 1 : (128 / G);
  constexpr int WBLK = 1024 + NG * 256;   // one n-tile's share of a ring stage: packed block + NG x 64 scale/zero words
  constexpr int WST = NT * WBLK;
  constexpr int ADMA = 4 * MT / WAVES;    // x DMAs per wave and stage (4 rows each)
  static_assert(ADMA >= 1, "every wave issues the same number of DMAs per stage");
  constexpr int ABUF = MT * 4096;         // one x stage: 16 MT rows x 256 B
  constexpr int NGD = (ABL == 6) ? 0 : NG;    // scale / zero DMAs per n-tile and k-block (ABL 6: none, timing only)
  constexpr int LPS = ADMA + NT * (1 + NGD);  // DMAs per wave and stage
  constexpr int SLOTS = 16 * NT;          // schedule slots per k-block (MT / 4 MFMAs each)
  constexpr int KW = rb_wstages(WAVES, MT, WST);  // weight ring stages
  constexpr int WDMAS = NT * (1 + NGD);   // weight DMAs per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][128 rows][256 B] x | [WAVES][KW][WST]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = PROD && wave_id >= WAVES;
  const int wave = producer ? wave_id - WAVES : wave_id;  // the consumer this wave is, or feeds
  const int nl = lane & 15, grp = lane >> 4;
  const int m0 = blockIdx.y * (16 * MT);
  const int ntiles = N >> 4;
  const int kblocks = K >> 7;
  const int tile0 = blockIdx.x * (WAVES * NT) + wave * NT;  // this wave's first n-tile
  const int S = gridDim.z, ks = blockIdx.z;
  const int kb0 = (int)(((long long)kblocks * ks) / S);
  const int nkb = (int)(((long long)kblocks * (ks + 1)) / S) - kb0;
  const s16x4 ident = identity_fragment(lane);

  // x DMA i of this wave fills rows 4 (ADMA wave + i) + (lane >> 4), chunk position lane & 15
  uint32_t aoff[ADMA];  // byte offsets from x + k * 128 (rows past M alias row M - 1; never stored)
#pragma unroll
  for (int i = 0; i < ADMA; ++i) {
    const int row = 4 * (ADMA * wave + i) + (lane >> 4);
    aoff[i] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)K * 2u + (((lane & 15) ^ (row & 15)) << 4);
  }
  const uint32_t a_lds = lds_offset(smem);
  const uint32_t w_lds = a_lds + kRbStages * ABUF + wave * (KW * WST);
  // One DMA of a k-block's LPS (compile-time index): the x rows first (into x stage `stage`, k-block kb), then per n-tile the
  // packed block and its scale/zero words (into weight stage `wstage`, k-block kbw).  k-blocks past the end are clamped: the fill
  // re-reads the last block (unused)
  auto issue_one = [&](auto idx_c, int stage, int kb, int wstage, int kbw) {
    constexpr int idx = decltype(idx_c)::value;
    if (ABL == 4) return;
    if constexpr (idx < ADMA) {
      const int k = kb0 + min(kb, nkb - 1);
      dma_b128_s(x + (size_t)k * 128, aoff[idx], a_lds + stage * ABUF + (ADMA * wave + idx) * 1024);
    } else {
      const int k = kb0 + min(kbw, nkb - 1);
      constexpr int t = (idx - ADMA) / (1 + NGD), part = (idx - ADMA) % (1 + NGD);
      const int tile = min(tile0 + t, ntiles - 1);  // tiles past N alias the last one; their columns are never stored
      const uint32_t dst = w_lds + wstage * WST + t * WBLK;
      if constexpr (part == 0) {
        dma_b128_nt_s(qdata + ((size_t)tile * kblocks + k) * 64, lane * 16, dst);
      } else {
        const int kg0 = (G >= 128) ? ((k * 128) / G) : (k * NG);
        dma_b32_s(sz + (size_t)(kg0 + part - 1) * N + tile * 16, nl * 4, dst + 1024 + (part - 1) * 256);
      }
    }
  };
  auto issue_x = [&](int stage, int kb) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (issue_one(std::integral_constant<int, I>{}, stage, kb, 0, 0), ...); }
    (std::make_integer_sequence<int, ADMA>{});
  };
  auto issue_w = [&](int wstage, int kbw) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (issue_one(std::integral_constant<int, ADMA + I>{}, 0, 0, wstage, kbw), ...); }
    (std::make_integer_sequence<int, WDMAS>{});
  };

  if (producer) {
    // same issue order and waits as the fused form below: w(0 .. KW-3) | x(0) w(KW-2) | x(1), then per k-block x(kb + 2), w(kb + KW - 1)
#pragma unroll
    for (int i = 0; i < KW - 2; ++i) issue_w(i, i);
    issue_x(0, 0); issue_w(KW - 2, KW - 2);
    issue_x(1, 1);
    int stage = 0, wstage = 0;
    for (int kb = 0; kb < nkb; ++kb) {
      if (kb < 2 || KW < 4) wait_vmcnt<LPS>(); else wait_vmcnt<LPS + WDMAS>();
      asm volatile("s_barrier" ::: "memory");  // stage kb has landed (every producer waited), the consumers are done with stage kb - 1
      issue_x((stage == 0) ? 2 : stage - 1, kb + 2);
      issue_w((wstage == 0) ? KW - 1 : wstage - 1, kb + KW - 1);
      stage = (stage == 2) ? 0 : stage + 1;
      wstage = (wstage == KW - 1) ? 0 : wstage + 1;
    }
    wait_vmcnt<0>();  // the clamped fills past the end still write LDS
    asm volatile("s_barrier" ::: "memory");
    if (S > 1) {
      f32x4 none[MT * NT];
      (void)split_k_meet<MT * NT, 64 * WAVES>(none, ws, tickets, blockIdx.y * gridDim.x + blockIdx.x, S, ks, tid, reinterpret_cast<int*>(smem), false);
    }
    return;
  }

  f32x4 acc[MT * NT];  // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < MT * NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane (row r = nl, group g = 2a + t): A operand of phase p = chunk 2p ^ (8a | t) of row r, at position chunk ^ r
  const int ga = grp >> 1, gt = grp & 1;
  const int pbase = nl * 256 + ((((ga << 3) | gt) ^ nl) << 4);  // ^ (p << 5) per phase, + 4096 per m-tile
  // its weight words: words {2a, 2a + 1} of packed lanes (n, kq = 2t) and (n, kq = 2t + 1)
  const int wbase = ((2 * gt) * 16 + nl) * 16 + 8 * ga;        // second piece: + 256
  // scale/zero word of word j = 2a + e
  const int zg0 = (G >= 128) ? 0 : (G == 64) ? ga : 2 * ga, zg1 = (G >= 128) ? 0 : (G == 64) ? ga : 2 * ga + 1;

  auto kblock = [&](int stage, int refill, int wstage, int wrefill, int kb) {
    const char* A = smem + stage * ABUF;
    const char* Wst = smem + kRbStages * ABUF + (wave * KW + wstage) * WST;
    uint32_t word[NT][2][2];  // [tile][e][which packed lane]
    float sc[NT][2], zp[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x2 pa = *reinterpret_cast<const u32x2*>(Wst + t * WBLK + wbase);
      const u32x2 pb = *reinterpret_cast<const u32x2*>(Wst + t * WBLK + wbase + 256);
      word[t][0][0] = pa.x; word[t][1][0] = pa.y; word[t][0][1] = pb.x; word[t][1][1] = pb.y;
      const uint32_t z0 = *reinterpret_cast<const uint32_t*>(Wst + t * WBLK + 1024 + zg0 * 256 + nl * 4);
      const uint32_t z1 = (G >= 64) ? z0 : *reinterpret_cast<const uint32_t*>(Wst + t * WBLK + 1024 + zg1 * 256 + nl * 4);
      sc[t][0] = bf16_lo_to_f32(z0); zp[t][0] = bf16_hi_to_f32(z0);
      sc[t][1] = bf16_lo_to_f32(z1); zp[t][1] = bf16_hi_to_f32(z1);
    }
    auto read_a = [&](u32x4 (&a)[MT], int p) {
      const char* ap = A + (pbase ^ (p << 5));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (ABL == 3) { a[mt] = u32x4{(uint32_t)pbase, (uint32_t)mt, (uint32_t)p, word[0][0][0]}; continue; }
        a[mt] = *reinterpret_cast<const u32x4*>(ap + mt * 4096);
      }
    };
    // dequant stage st of word (tile t, e, which)
    auto stage_of = [&](auto st_c, DequantPipe& d, int t, int e, int which) {
      if (ABL == 2) {
        if constexpr (decltype(st_c)::value == 3) { d.out[0] = word[t][e][which]; d.out[1] = d.out[0] + 1; d.out[2] = d.out[0] ^ 5; d.out[3] = d.out[0] + 7; }
        return;
      }
      dequant_stage<decltype(st_c)::value>(d, word[t][e][which], sc[t][e], -8.0f * sc[t][e], zp[t][e], ident);
    };
    DequantPipe dq[NT][2];
    u32x4 a[MT], an[MT];
    u32x4 bv[NT][2];  // [tile][h] B operands of the current e
    auto take_b = [&] {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bv[t][0] = u32x4{dq[t][0].out[0], dq[t][0].out[1], dq[t][1].out[0], dq[t][1].out[1]};
        bv[t][1] = u32x4{dq[t][0].out[2], dq[t][0].out[3], dq[t][1].out[2], dq[t][1].out[3]};
      }
    };
    read_a(a, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int wh = 0; wh < 2; ++wh) {
        stage_of(std::integral_constant<int, 0>{}, dq[t][wh], t, 0, wh);
        stage_of(std::integral_constant<int, 1>{}, dq[t][wh], t, 0, wh);
        stage_of(std::integral_constant<int, 2>{}, dq[t][wh], t, 0, wh);
        stage_of(std::integral_constant<int, 3>{}, dq[t][wh], t, 0, wh);
      }
    take_b();
    [&]<int... P>(std::integer_sequence<int, P...>) {
      (([&] {
         constexpr int p = P, e = p >> 1, h = p & 1;
         if (p < 3) read_a(an, p + 1);
         [&]<int... C>(std::integer_sequence<int, C...>) {
           (([&] {
              constexpr int c = C, t = c / 4, slot = p * 4 * NT + c;
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (slot < LPS && !PROD) issue_one(std::integral_constant<int, slot>{}, refill, kb + 2, wrefill, kb + KW - 1);
              if constexpr (e == 0) {  // the 8 NT slots of phases 0, 1 carry the 2 NT words of e = 1, four stages each
                constexpr int q = h * 4 * NT + c, w = q / 4;
                stage_of(std::integral_constant<int, q % 4>{}, dq[w / 2][w % 2], w / 2, 1, w % 2);
              }
#pragma unroll
              for (int mt = (c % 4) * MT / 4; mt < (c % 4 + 1) * MT / 4; ++mt) {  // this slot's share of the tile's MT MFMAs
                if (ABL == 1) { acc[t * MT + mt].x += bits_to_f32(a[mt].x ^ bv[t][h].x ^ a[mt].y ^ a[mt].z ^ a[mt].w ^ bv[t][h].y ^ bv[t][h].z ^ bv[t][h].w); continue; }
                acc[t * MT + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[mt]),
                                                                          __builtin_bit_cast(bf16x8, bv[t][h]), acc[t * MT + mt], 0, 0, 0);
              }
            }()),
            ...);
         }(std::make_integer_sequence<int, 4 * NT>{});
         __builtin_amdgcn_sched_barrier(0);
         if constexpr (p == 1) take_b();
         if constexpr (p < 3) {
#pragma unroll
           for (int mt = 0; mt < MT; ++mt) a[mt] = an[mt];
         }
       }()),
       ...);
    }(std::make_integer_sequence<int, 4>{});
    // DMAs left over when a stage has more of them than the k-block has slots
    if constexpr (LPS > SLOTS && !PROD) {
      [&]<int... I>(std::integer_sequence<int, I...>) { (issue_one(std::integral_constant<int, SLOTS + I>{}, refill, kb + 2, wrefill, kb + KW - 1), ...); }
      (std::make_integer_sequence<int, LPS - SLOTS>{});
    }
  };

  // issue order: w(0 .. KW-3) | x(0) w(KW-2) | x(1) ... then per k-block kb: x(kb + 2), w(kb + KW - 1).  With KW >= 4, when k-block kb
  // starts the requests younger than x(kb) are w(kb + KW - 3) (issued right behind it), x(kb + 1) and w(kb + KW - 2): LPS + WDMAS of
  // them may still be in flight; everything this k-block reads -- x(kb), w(kb) -- is older and has landed.
  if constexpr (!PROD) {
#pragma unroll
    for (int i = 0; i < KW - 2; ++i) issue_w(i, i);
    issue_x(0, 0); issue_w(KW - 2, KW - 2);
    issue_x(1, 1);
  }
  if (ABL == 5) ts[1] = __builtin_amdgcn_s_memtime();
  int stage = 0, wstage = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    // (k-blocks 0 and 1: x(kb + 1) follows x(kb) directly or with one weight stage between -- only LPS younger requests exist)
    // (KW = 3: w(kb) is issued right BEHIND x(kb), in k-block kb - 2, so only x(kb + 1) and w(kb + 1) -- LPS requests -- are younger
    // than what this k-block reads; a deeper weight ring issues w(kb) earlier and one more weight stage may be in flight)
    if (ABL != 4 && !PROD) { if (kb < 2 || KW < 4) wait_vmcnt<LPS>(); else wait_vmcnt<LPS + WDMAS>(); }
    // everyone's share has landed, and everyone has finished reading stage kb - 1 (its LDS reads have returned)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ABL == 5 && kb < 8) ts[2 + kb] = __builtin_amdgcn_s_memtime();
    const int refill = (stage == 0) ? 2 : stage - 1;          // (kb + 2) % 3 == (kb - 1) % 3
    const int wrefill = (wstage == 0) ? KW - 1 : wstage - 1;  // (kb + KW - 1) % KW == (kb - 1) % KW
    kblock(stage, refill, wstage, wrefill, kb);
    stage = (stage == 2) ? 0 : stage + 1;
    wstage = (wstage == KW - 1) ? 0 : wstage + 1;
  }
  if constexpr (!PROD) wait_vmcnt<0>();  // the clamped fills past the end still write LDS
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (ABL == 5) ts[10] = __builtin_amdgcn_s_memtime();
  auto dump = [&] {
    if (ABL == 5 && trace != nullptr && tid == 0) {
      ts[12] = __builtin_amdgcn_s_memtime();
      unsigned long long* t = trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
      for (int i = 0; i < 13; ++i) t[i] = ts[i];
    }
  };

  if (S > 1 && !split_k_meet<MT * NT, 64 * WAVES>(acc, ws, tickets, blockIdx.y * gridDim.x + blockIdx.x, S, ks, tid, reinterpret_cast<int*>(smem))) {
    dump();
    return;
  }
  if (ABL == 5) ts[11] = __builtin_amdgcn_s_memtime();

  // D layout of the 16x16 tile: lane (col = nl, group g) holds rows 4 g + {0..3}
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (tile0 + t >= ntiles) continue;
    const int nn = (tile0 + t) * 16 + nl;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + grp * 4 + r;
        if (m < M) y[(size_t)m * N + nn] = f32_to_bf16_bits(acc[t * MT + mt][r]);
      }
  }
  dump();
}

thread_local unsigned long long* g_mm_trace = nullptr;  // profiling only (ao_int4_set_trace)
thread_local int g_tune_wpb = 0;
// 128 x 256 tiles (64-column wave tiles) of the batched kernel.  Round 6, first fit (Llama-3-8B shapes): M >= 512 and tiles that fill >= 7/8 of every
// round of the 256 CUs.  Re-fitted on 16 shapes of four models x M = 256 / 512 / 1024 x 7 forms (profiles/int4_forms_big_r06.jsonl): within ONE round
// they are ahead from 144 tiles on (qkv 6144 x 4096 at M = 1024, 192 tiles: 70.3 -> 61.3 us; 5120^2, 160: 84.9 -> 71.6; 4608 x 3584, 144: 60.7 -> 52.9;
// 18944 x 3584 at M = 256, 148: 64.0 -> 57.5; 10240 x 8192 at 512, 160: 131 -> 121) and behind up to 128 (8192^2 at 512: 75.7 -> 88.4); over several
// rounds from 0.8 of full (13824 x 5120 at 1024, 432 tiles: 193 -> 174; at 0.58 - 0.63: 10 - 13 % behind).
inline bool int4_mm_w32_tiles64(int64_t M, int64_t N) {
  const int64_t tiles = ((N + 255) / 256) * ((M + 127) / 128), rounds = (tiles + 255) / 256;
  return M >= 256 && ((rounds == 1 && tiles >= 144) || (tiles >= 200 && tiles * 10 >= rounds * 256 * 8));
}

// ---------------------------------------------------------------------------
// Round 5: int4_mm_w32_kernel -- the batched kernel on 128 x 128 tiles with v_mfma_f32_32x32x16_bf16.
//
// int4_mm_rb_kernel above multiplies 16 x 16 x 32: per k-block and n-tile a wave issues 32 MFMAs and reads 32 A fragments (one
// ds_read_b128 each) -- at one consumer wave per SIMD the k-block is the SUM of its MFMA issue (512 cycles), its fragment reads (~512)
// and its exact dequant (~300), DESIGN.md 4.3b.  The 32 x 32 x 16 instruction does twice the work per issue and per A fragment:
//   * a wave owns TWO n-tiles (32 columns) and all 128 rows: lane (n = lane & 31, g = lane >> 5) -- tile t = n >> 4, nl = n & 15 --
//     reads the 16 bytes (words j = 0 .. 3) of packed lanes (nl, kq = 2 g) and (nl, 2 g + 1) of its tile's block: the 8 words whose
//     runs k = 32 j + 16 h + 8 g + {0 .. 7} are contiguous per (j, h) -- one 16-byte chunk of x, chunk 4 j + 2 h + g.  MFMA step
//     s = 2 j + h (8 per k-block) multiplies the two lane groups' chunks 2 s + g: B = the 8 dequantised weights (word j, half h of
//     both packed lanes), A = chunk 2 s + g of the lane's row of each 32-row m-tile -- ONE ds_read_b128;
//   * per k-block and wave: 32 MFMAs of 32 cycles for 32 columns (was 64 of 16), 32 fragment reads (was 64), the same 8 words of
//     exact dequant -- whose four stages per word ride in the MFMAs' issue shadow, one stage per MFMA, a word pair ahead;
//   * workgroup = 4 such waves (+ 4 DMA-producer waves): 128 x 128 outputs; x tile and weight rings exactly as above (same LDS-DMA
//     layouts, same swizzle, same waits); narrow weights cut K and meet through splitk.h.
// Same products, fp32 accumulation in a different order than the 16 x 16 kernel: <= 1e-3 of the oracle like it.
// ---------------------------------------------------------------------------
template <int G, bool PROD, int ABL = 0, int CG = 1>
__global__ __launch_bounds__(PROD ? 512 : 256) void int4_mm_w32_kernel(
    const uint16_t* __restrict__ x, const u32x4* __restrict__ qdata, const uint32_t* __restrict__ sz, uint16_t* __restrict__ y, int M,
    int N, int K, float* __restrict__ ws, unsigned* __restrict__ tickets, unsigned long long* __restrict__ trace) {
  // CG (round 6): 32-column groups per wave.  2: a wave owns 64 columns -- every A fragment feeds TWO MFMAs (one per group): half the LDS
  // fragment reads per MFMA, 128 accumulator registers, a 128 x 256 workgroup tile (G >= 128 only: the weight rings of four 16-column tiles
  // per wave and the 96 KiB x ring fill the LDS)
  constexpr int WAVES = 4, NT = 2 * CG;
  constexpr int NG = (G >= 128) ? 1 : (128 / G);
  constexpr int WBLK = 1024 + NG * 256;
  constexpr int WST = NT * WBLK;
  constexpr int ADMA = 8;                  // x DMAs per wave and stage (4 rows each: 128 rows / 4 waves)
  constexpr int ABUF = 128 * 256;          // one x stage
  constexpr int WDMAS = NT * (1 + NG);
  constexpr int LPS = ADMA + WDMAS;
  constexpr int KW = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][128 rows][256 B] x | [4 waves][3][WST]
  unsigned long long ts[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (ABL == 5) ts[0] = __builtin_amdgcn_s_memtime();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = PROD && wave_id >= WAVES;
  const int wave = producer ? wave_id - WAVES : wave_id;
  const int nl = lane & 15, lt = (lane >> 4) & 1, g = lane >> 5;
  const int m0 = blockIdx.y * 128;
  const int ntiles = N >> 4;
  const int kblocks = K >> 7;
  const int tile0 = blockIdx.x * (WAVES * NT) + wave * NT;
  const int S = gridDim.z, ks = blockIdx.z;
  const int kb0 = (int)(((long long)kblocks * ks) / S);
  const int nkb = (int)(((long long)kblocks * (ks + 1)) / S) - kb0;
  const s16x4 ident = identity_fragment(lane);

  uint32_t aoff[ADMA];
#pragma unroll
  for (int i = 0; i < ADMA; ++i) {
    const int row = 4 * (ADMA * wave + i) + (lane >> 4);
    aoff[i] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)K * 2u + (((lane & 15) ^ (row & 15)) << 4);
  }
  const uint32_t a_lds = lds_offset(smem);
  const uint32_t w_lds = a_lds + 3 * ABUF + wave * (KW * WST);
  auto issue_one = [&](auto idx_c, int stage, int kb, int wstage, int kbw) {
    constexpr int idx = decltype(idx_c)::value;
    if constexpr (idx < ADMA) {
      const int k = kb0 + min(kb, nkb - 1);
      dma_b128_s(x + (size_t)k * 128, aoff[idx], a_lds + stage * ABUF + (ADMA * wave + idx) * 1024);
    } else {
      const int k = kb0 + min(kbw, nkb - 1);
      constexpr int t = (idx - ADMA) / (1 + NG), part = (idx - ADMA) % (1 + NG);
      const int tile = min(tile0 + t, ntiles - 1);
      const uint32_t dst = w_lds + wstage * WST + t * WBLK;
      if constexpr (part == 0) {
        dma_b128_nt_s(qdata + ((size_t)tile * kblocks + k) * 64, lane * 16, dst);
      } else {
        const int kg0 = (G >= 128) ? ((k * 128) / G) : (k * NG);
        dma_b32_s(sz + (size_t)(kg0 + part - 1) * N + tile * 16, nl * 4, dst + 1024 + (part - 1) * 256);
      }
    }
  };
  auto issue_x = [&](int stage, int kb) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (issue_one(std::integral_constant<int, I>{}, stage, kb, 0, 0), ...); }
    (std::make_integer_sequence<int, ADMA>{});
  };
  auto issue_w = [&](int wstage, int kbw) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (issue_one(std::integral_constant<int, ADMA + I>{}, 0, 0, wstage, kbw), ...); }
    (std::make_integer_sequence<int, WDMAS>{});
  };

  if (producer) {
    // issue order and waits of int4_mm_rb_kernel's producers (KW = 3): w(0) | x(0) w(1) | x(1), then per k-block x(kb + 2), w(kb + 2)
    issue_w(0, 0);
    issue_x(0, 0); issue_w(1, 1);
    issue_x(1, 1);
    int stage = 0;
    for (int kb = 0; kb < nkb; ++kb) {
      wait_vmcnt<LPS>();
      asm volatile("s_barrier" ::: "memory");
      const int refill = (stage == 0) ? 2 : stage - 1;
      issue_x(refill, kb + 2);
      issue_w(refill, kb + 2);
      stage = (stage == 2) ? 0 : stage + 1;
    }
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    if (S > 1) {
      f32x4 none[16 * CG];
      (void)split_k_meet<16 * CG, 64 * WAVES>(none, ws, tickets, blockIdx.y * gridDim.x + blockIdx.x, S, ks, tid, reinterpret_cast<int*>(smem), false);
    }
    return;
  }

  f32x16 acc[4 * CG];
#pragma unroll
  for (int i = 0; i < 4 * CG; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // A fragment of step s, m-tile mt: chunk 2 s + g of row 32 mt + (lane & 31), at position chunk ^ (row & 15) = chunk ^ nl
  const int arow = (lane & 31) * 256;
  // the lane's packed words: 16 bytes of packed lanes (nl, 2 g) and (nl, 2 g + 1) of tile lt
  const int wbase = lt * WBLK + ((2 * g) * 16 + nl) * 16;  // (+ 2 c WBLK for column group c)

  auto kblock = [&](int stage, int refill, int kb) {
    const char* A = smem + stage * ABUF;
    const char* Wst = smem + 3 * ABUF + (wave * KW + stage) * WST;
    u32x4 pa[CG], pb[CG];
    float sc[CG][NG], zp[CG][NG];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      pa[c] = *reinterpret_cast<const u32x4*>(Wst + 2 * c * WBLK + wbase);
      pb[c] = *reinterpret_cast<const u32x4*>(Wst + 2 * c * WBLK + wbase + 256);
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const uint32_t z = *reinterpret_cast<const uint32_t*>(Wst + (2 * c + lt) * WBLK + 1024 + q * 256 + nl * 4);
        sc[c][q] = bf16_lo_to_f32(z); zp[c][q] = bf16_hi_to_f32(z);
      }
    }
    constexpr auto group_of = [](int j) constexpr { return (G >= 128) ? 0 : (G == 64) ? (j >> 1) : j; };
    DequantPipe dq[CG][2][2];  // [column group][j & 1][which]: word j is dequantised while word j - 1 multiplies
    // ABL (laboratory builds only, wrong results): 1 no 32 x 32 x 16 MFMAs, 2 no dequant, 3 no A-fragment reads; 5 = product + s_memtime stamps
    auto word_of = [&](int c, int j, int which) -> uint32_t {
      const u32x4& v = which ? pb[c] : pa[c];
      return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
    };
    auto stage_of = [&](auto st_c, auto j_c, auto which_c, auto c_c) {  // (compile-time indices: everything stays in registers)
      constexpr int j = decltype(j_c)::value, which = decltype(which_c)::value, q = group_of(j), c = decltype(c_c)::value;
      if (ABL == 2) {
        if constexpr (decltype(st_c)::value == 3) { DequantPipe& d = dq[c][j & 1][which]; d.out[0] = word_of(c, j, which); d.out[1] = d.out[0] + 1; d.out[2] = d.out[0] ^ 5; d.out[3] = d.out[0] + 7; }
        return;
      }
      dequant_stage<decltype(st_c)::value>(dq[c][j & 1][which], word_of(c, j, which), sc[c][q], -8.0f * sc[c][q], zp[c][q], ident);
    };
    auto read_a = [&](int s, int mt) {
      if (ABL == 3) return u32x4{(uint32_t)s, (uint32_t)mt, pa[0].x, pb[0].y};
      return *reinterpret_cast<const u32x4*>(A + mt * (32 * 256) + arow + ((((2 * s + g) ^ nl) & 15) << 4));
    };
    // word 0 first (nothing to hide it behind: its data landed with this k-block's barrier)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    [&]<int... C>(std::integer_sequence<int, C...>) {
      ((stage_of(I0{}, I0{}, I0{}, std::integral_constant<int, C>{}), stage_of(I1{}, I0{}, I0{}, std::integral_constant<int, C>{}),
        stage_of(I2{}, I0{}, I0{}, std::integral_constant<int, C>{}), stage_of(I3{}, I0{}, I0{}, std::integral_constant<int, C>{}),
        stage_of(I0{}, I0{}, I1{}, std::integral_constant<int, C>{}), stage_of(I1{}, I0{}, I1{}, std::integral_constant<int, C>{}),
        stage_of(I2{}, I0{}, I1{}, std::integral_constant<int, C>{}), stage_of(I3{}, I0{}, I1{}, std::integral_constant<int, C>{})),
       ...);
    }(std::make_integer_sequence<int, CG>{});
    // A fragments are requested AHD fragments ahead (an LDS round trip is ~100 cycles, an MFMA 32): a ring of AHD + 1 register quads
    constexpr int AHD = 3;
    u32x4 aq[AHD + 1];
#pragma unroll
    for (int i = 0; i < AHD; ++i) aq[i] = read_a(i >> 2, i & 3);
    [&]<int... SL>(std::integer_sequence<int, SL...>) {
      (([&] {
         // slot = (fragment f = (step s, m-tile mt), column group c): a fragment is read once and multiplies CG times
         constexpr int sl = SL, f = sl / CG, c = sl % CG, s = f >> 2, mt = f & 3, j = s >> 1, h = s & 1;
         __builtin_amdgcn_sched_barrier(0);
         if constexpr (c == 0 && f + AHD < 32) aq[(f + AHD) % (AHD + 1)] = read_a((f + AHD) >> 2, (f + AHD) & 3);
         const u32x4 a_cur = aq[f % (AHD + 1)];
         if constexpr (!PROD && sl < LPS) issue_one(std::integral_constant<int, sl>{}, refill, kb + 2, refill, kb + 2);
         if constexpr (j < 3) {  // the 8 CG slots of word j carry the 8 CG stage calls of word j + 1: slot (h, mt, c) -> group c, which = h, stage = mt
           stage_of(std::integral_constant<int, mt>{}, std::integral_constant<int, (j < 3 ? j + 1 : 3)>{}, std::integral_constant<int, h>{}, std::integral_constant<int, c>{});
         }
         const DequantPipe& d0 = dq[c][j & 1][0];
         const DequantPipe& d1 = dq[c][j & 1][1];
         const u32x4 bv = {d0.out[2 * h], d0.out[2 * h + 1], d1.out[2 * h], d1.out[2 * h + 1]};
         if (ABL == 1) acc[c * 4 + mt][0] += bits_to_f32(a_cur.x ^ bv.x ^ a_cur.y ^ a_cur.z ^ a_cur.w ^ bv.y ^ bv.z ^ bv.w);
         else acc[c * 4 + mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur), __builtin_bit_cast(bf16x8, bv), acc[c * 4 + mt], 0, 0, 0);
       }()),
       ...);
    }(std::make_integer_sequence<int, 32 * CG>{});
    __builtin_amdgcn_sched_barrier(0);
  };

  if constexpr (!PROD) {
    issue_w(0, 0);
    issue_x(0, 0); issue_w(1, 1);
    issue_x(1, 1);
  }
  if (ABL == 5) ts[1] = __builtin_amdgcn_s_memtime();
  int stage = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    if constexpr (!PROD) wait_vmcnt<LPS>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ABL == 5 && kb < 8) ts[2 + kb] = __builtin_amdgcn_s_memtime();
    kblock(stage, (stage == 0) ? 2 : stage - 1, kb);
    stage = (stage == 2) ? 0 : stage + 1;
  }
  if constexpr (!PROD) wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (ABL == 5) ts[10] = __builtin_amdgcn_s_memtime();
  auto dump = [&] {
    if (ABL == 5 && trace != nullptr && tid == 0) {
      ts[12] = __builtin_amdgcn_s_memtime();
      unsigned long long* t = trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
      for (int i = 0; i < 13; ++i) t[i] = ts[i];
    }
  };

  if (S > 1) {
    f32x4 part[16 * CG];
#pragma unroll
    for (int i = 0; i < 16 * CG; ++i) part[i] = f32x4{acc[i >> 2][4 * (i & 3)], acc[i >> 2][4 * (i & 3) + 1], acc[i >> 2][4 * (i & 3) + 2], acc[i >> 2][4 * (i & 3) + 3]};
    if (!split_k_meet<16 * CG, 64 * WAVES>(part, ws, tickets, blockIdx.y * gridDim.x + blockIdx.x, S, ks, tid, reinterpret_cast<int*>(smem))) {
      dump();
      return;
    }
#pragma unroll
    for (int i = 0; i < 16 * CG; ++i) {
      acc[i >> 2][4 * (i & 3)] = part[i].x; acc[i >> 2][4 * (i & 3) + 1] = part[i].y;
      acc[i >> 2][4 * (i & 3) + 2] = part[i].z; acc[i >> 2][4 * (i & 3) + 3] = part[i].w;
    }
  }
  if (ABL == 5) ts[11] = __builtin_amdgcn_s_memtime();

  // D layout of the 32 x 32 tile: lane (col = lane & 31, g) holds rows (r & 3) + 8 (r >> 2) + 4 g, r = 0 .. 15.  Through the idle LDS
  // and out as 16-byte row pieces (the direct form would be 64 two-byte stores per lane).
  {
    constexpr int BN = 128 * CG;      // columns of the workgroup tile
    constexpr int RS = BN * 2 + 16;   // staging row stride in bytes
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      const int col = wave * (32 * CG) + c * 32 + (lane & 31);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * g;
          *reinterpret_cast<uint16_t*>(smem + row * RS + col * 2) = f32_to_bf16_bits(acc[c * 4 + mt][r]);
        }
    }
    // (only the consumer waves are here: the producers left after the loop's last barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // NOTE: see the launch: the barrier counts the waves still alive
#pragma unroll
    for (int it = 0; it < (128 * (BN / 8)) / 256; ++it) {
      const int cc = it * 256 + tid;
      const int row = cc / (BN / 8), piece = cc % (BN / 8);
      const int m = m0 + row, n = blockIdx.x * BN + piece * 8;
      if (m < M && n + 8 <= N)
        *reinterpret_cast<u32x4*>(y + (size_t)m * N + n) = *reinterpret_cast<const u32x4*>(smem + row * RS + piece * 16);
    }
  }
  dump();
}

template <int G, bool PROD, int ABL = 0, int CG = 1>
int launch_mm_w32(const uint16_t* x, const int32_t* qdata, const uint16_t* sz, uint16_t* y, int64_t M, int64_t N, int64_t K, int split,
                  hipStream_t stream) {
  constexpr int NG = (G >= 128) ? 1 : (128 / G);
  constexpr int BN = 128 * CG;
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + 127) / 128), (unsigned)split), block(PROD ? 512 : 256);
  constexpr size_t smem = (size_t)3 * 128 * 256 + (size_t)4 * 3 * 2 * CG * (1024 + NG * 256);
  static_assert(smem <= 160 * 1024, "int4_mm_w32_kernel: LDS");
  static_assert((size_t)128 * (BN * 2 + 16) <= smem, "int4_mm_w32_kernel: the epilogue's staging tile");
  float* ws = nullptr;
  unsigned* tickets = nullptr;
  if (split > 1) {
    AO_REQUIRE((int64_t)grid.x * grid.y * split * CG <= kSplitMaxTiles, "int4_mm_w32: %u x %u tiles x %d parts exceed the split-K workspace", grid.x, grid.y, split);
    AO_REQUIRE((int64_t)grid.x * grid.y <= kSplitMaxTickets - 8, "int4_mm_w32: %u x %u output tiles exceed the split-K tickets", grid.x, grid.y);
    if (int rc = splitk_workspace(stream, &ws, &tickets, (size_t)grid.x * grid.y * split * 128 * BN, split)) return rc;
  }
  auto kern = int4_mm_w32_kernel<G, PROD, ABL, CG>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(int4_mm_w32_kernel)")) return rc;
  ao::launch(kern, grid, block, smem, stream, x, reinterpret_cast<const u32x4*>(qdata), reinterpret_cast<const uint32_t*>(sz), y, (int)M,
             (int)N, (int)K, ws, tickets, g_mm_trace);
  AO_LAUNCH_CHECK("int4_mm_w32_kernel launch");
  return AO_OK;
}


// (round 5: int4_mm_kh_kernel -- round 3's K-half-wave form, modes 84S / 85S / 86x, measured 3 % behind the product and never dispatched --
// was removed; DESIGN.md 4.3b keeps its measurements, git history its source.)

template <int G, int WAVES, int NT, int MT = 8, int ABL = 0, bool PROD = false>
int launch_mm_rb(const uint16_t* x, const int32_t* qdata, const uint16_t* sz, uint16_t* y, int64_t M, int64_t N, int64_t K, int split,
                 hipStream_t stream) {
  constexpr int NG = (G >= 128) ? 1 : (128 / G);
  constexpr int BN = WAVES * NT * 16;
  constexpr int BM = 16 * MT;
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)split), block(64 * WAVES * (PROD ? 2 : 1));
  constexpr int KW = rb_wstages(WAVES, MT, NT * (1024 + NG * 256));
  constexpr size_t smem = (size_t)kRbStages * MT * 4096 + (size_t)WAVES * KW * NT * (1024 + NG * 256);
  static_assert(smem <= 160 * 1024, "int4_mm_rb_kernel: LDS");
  float* ws = nullptr;
  unsigned* tickets = nullptr;
  if (split > 1) {
    AO_REQUIRE((int64_t)grid.x * grid.y * split * BN * BM <= (int64_t)kSplitMaxTiles * 128 * 128, "int4_mm_rb: %u x %u tiles x %d parts exceed the split-K workspace",
               grid.x, grid.y, split);
    AO_REQUIRE((int64_t)grid.x * grid.y <= kSplitMaxTickets, "int4_mm_rb: %u x %u output tiles exceed the split-K tickets", grid.x, grid.y);
    if (int rc = splitk_workspace(stream, &ws, &tickets, (size_t)grid.x * grid.y * split * BN * BM, split)) return rc;
  }
  auto kern = int4_mm_rb_kernel<G, WAVES, NT, MT, ABL, PROD>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(int4_mm_rb_kernel)")) return rc;
  ao::launch(kern, grid, block, smem, stream, x, reinterpret_cast<const u32x4*>(qdata), reinterpret_cast<const uint32_t*>(sz), y, (int)M,
             (int)N, (int)K, ws, tickets, g_mm_trace);
  AO_LAUNCH_CHECK("int4_mm_rb_kernel launch");
  return AO_OK;
}

// ---------------------------------------------------------------------------
// pack / unpack / dequantize / fused quantize
// ---------------------------------------------------------------------------

// one thread per output word: (ntile16, ksuper, t, j)
__global__ void int4_pack_kernel(const uint8_t* __restrict__ w_u8, uint32_t* __restrict__ out,
                                 int64_t N, int64_t K, int64_t total_words) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_words) return;
  const int j = idx & 3;
  const int t = (idx >> 2) & 63;
  const int64_t blk = idx >> 8;  // ntile * ksuper_count + ksuper
  const int64_t ks_count = K >> 7;
  const int64_t nt = blk / ks_count, ks = blk % ks_count;
  const int64_t n = nt * 16 + (t & 15);
  const uint8_t* row = w_u8 + n * (K >> 1);
  uint32_t v[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t k0 = (ks * 8 + 2 * j + h) * 16 + (t >> 4) * 4;  // multiple of 4
    const uint8_t b0 = row[k0 >> 1], b1 = row[(k0 >> 1) + 1];
    v[4 * h + 0] = b0 >> 4; v[4 * h + 1] = b0 & 0xF;  // even k in the high nibble
    v[4 * h + 2] = b1 >> 4; v[4 * h + 3] = b1 & 0xF;
  }
  out[idx] = v[0] | (v[2] << 4) | (v[4] << 8) | (v[6] << 12) | (v[1] << 16) | (v[3] << 20) |
             (v[5] << 24) | (v[7] << 28);
}

__global__ void int4_unpack_kernel(const uint32_t* __restrict__ qdata, uint8_t* __restrict__ w_u8,
                                   int64_t N, int64_t K, int64_t total_words) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_words) return;
  const int j = idx & 3;
  const int t = (idx >> 2) & 63;
  const int64_t blk = idx >> 8;
  const int64_t ks_count = K >> 7;
  const int64_t nt = blk / ks_count, ks = blk % ks_count;
  const int64_t n = nt * 16 + (t & 15);
  const uint32_t p = qdata[idx];
  uint32_t v[8];
  v[0] = p & 0xF; v[2] = (p >> 4) & 0xF; v[4] = (p >> 8) & 0xF; v[6] = (p >> 12) & 0xF;
  v[1] = (p >> 16) & 0xF; v[3] = (p >> 20) & 0xF; v[5] = (p >> 24) & 0xF; v[7] = (p >> 28) & 0xF;
  uint8_t* row = w_u8 + n * (K >> 1);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t k0 = (ks * 8 + 2 * j + h) * 16 + (t >> 4) * 4;
    row[k0 >> 1] = (uint8_t)((v[4 * h + 0] << 4) | v[4 * h + 1]);
    row[(k0 >> 1) + 1] = (uint8_t)((v[4 * h + 2] << 4) | v[4 * h + 3]);
  }
}

// one thread per packed word -> 8 bf16 of the [N][K] dequantised weight
template <int G>
__global__ void int4_dequant_kernel(const uint32_t* __restrict__ qdata,
                                    const uint32_t* __restrict__ sz, uint16_t* __restrict__ w,
                                    int64_t N, int64_t K, int64_t total_words) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_words) return;
  const int j = idx & 3;
  const int t = (idx >> 2) & 63;
  const int64_t blk = idx >> 8;
  const int64_t ks_count = K >> 7;
  const int64_t nt = blk / ks_count, ks = blk % ks_count;
  const int64_t n = nt * 16 + (t & 15);
  const int64_t kbase = ks * 128 + (2 * j) * 16 + (t >> 4) * 4;
  const uint32_t szv = sz[(kbase / G) * N + n];  // both inner tiles share the group (G >= 32)
  const float s = bf16_lo_to_f32(szv), z = bf16_hi_to_f32(szv);
  uint32_t b[4];
  dequant_word(qdata[idx], s, -8.0f * s, z, b);
  u32x2* d0 = reinterpret_cast<u32x2*>(w + n * K + kbase);
  u32x2* d1 = reinterpret_cast<u32x2*>(w + n * K + kbase + 16);
  *d0 = u32x2{b[0], b[1]};
  *d1 = u32x2{b[2], b[3]};
}

// Fused choose_qparams + quantize + tile-pack + scale/zero pack.
// One wave per (ntile16, span of max(128, G) k).  All arithmetic replays the
// reference's bf16 op sequence (each op: fp32 compute, RNE to bf16).
template <int G>
__global__ __launch_bounds__(64) void int4_quantize_kernel(const uint16_t* __restrict__ w,
                                                           u32x4* __restrict__ qdata,
                                                           uint32_t* __restrict__ sz, int64_t N,
                                                           int64_t K) {
  constexpr int KB_PER = (G > 128) ? (G / 128) : 1;  // 128-k blocks per wave
  constexpr int NG = (G >= 128) ? 1 : (128 / G);     // groups per 128-k block
  const int lane = threadIdx.x;
  const int64_t kblocks = K >> 7;
  const int64_t span = blockIdx.x;  // ntile * (kblocks / KB_PER) + s
  const int64_t spans_per_tile = kblocks / KB_PER;
  const int64_t nt = span / spans_per_tile;
  const int64_t kb_first = (span % spans_per_tile) * KB_PER;
  const int64_t n = nt * 16 + (lane & 15);
  const int kq = lane >> 4;

  float v[KB_PER][8][4];
  float gmin[KB_PER * NG], gmax[KB_PER * NG];
#pragma unroll
  for (int i = 0; i < KB_PER * NG; ++i) { gmin[i] = INFINITY; gmax[i] = -INFINITY; }

#pragma unroll
  for (int b = 0; b < KB_PER; ++b) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const u32x2 raw = *reinterpret_cast<const u32x2*>(w + n * K + (kb_first + b) * 128 + t * 16 + kq * 4);
      v[b][t][0] = bf16_lo_to_f32(raw.x); v[b][t][1] = bf16_hi_to_f32(raw.x);
      v[b][t][2] = bf16_lo_to_f32(raw.y); v[b][t][3] = bf16_hi_to_f32(raw.y);
      const int gi = (G >= 128) ? 0 : (b * NG + (t * 16) / G);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gmin[gi] = fminf(gmin[gi], v[b][t][i]);
        gmax[gi] = fmaxf(gmax[gi], v[b][t][i]);
      }
    }
  }
  // reduce over the 4 lanes (kq) that share n
#pragma unroll
  for (int i = 0; i < KB_PER * NG; ++i) {
    if (G >= 128 && i > 0) break;
    gmin[i] = fminf(gmin[i], __shfl_xor(gmin[i], 16)); gmin[i] = fminf(gmin[i], __shfl_xor(gmin[i], 32));
    gmax[i] = fmaxf(gmax[i], __shfl_xor(gmax[i], 16)); gmax[i] = fmaxf(gmax[i], __shfl_xor(gmax[i], 32));
  }

  constexpr int GROUPS = (G >= 128) ? 1 : NG;
  float sc[GROUPS], zp[GROUPS], minv[GROUPS];
#pragma unroll
  for (int i = 0; i < GROUPS; ++i) {
    float s = round_bf16(round_bf16(gmax[i] - gmin[i]) / 15.0f);
    s = fmaxf(s, 1.1754943508222875e-38f);  // bf16 smallest normal (default eps)
    const float s8 = round_bf16(s * 8.0f);
    const float z = round_bf16(gmin[i] + s8);
    sc[i] = s; zp[i] = z;
    minv[i] = round_bf16(z - s8);
    const int64_t kg = (G >= 128) ? ((kb_first * 128) / G) : (kb_first * NG + i);
    if (kq == 0) sz[kg * N + n] = (uint32_t)f32_to_bf16_bits(s) | ((uint32_t)f32_to_bf16_bits(z) << 16);
  }

#pragma unroll
  for (int b = 0; b < KB_PER; ++b) {
    uint32_t words[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t q[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = 2 * j + h;
        const int gi = (G >= 128) ? 0 : ((t * 16) / G);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = round_bf16(round_bf16(v[b][t][i] - minv[gi]) / sc[gi]);
          const float r = fminf(fmaxf(rintf(d), 0.f), 15.f);
          q[4 * h + i] = (uint32_t)(int)r;
        }
      }
      words[j] = q[0] | (q[2] << 4) | (q[4] << 8) | (q[6] << 12) | (q[1] << 16) | (q[3] << 20) |
                 (q[5] << 24) | (q[7] << 28);
    }
    qdata[(nt * kblocks + kb_first + b) * 64 + lane] = u32x4{words[0], words[1], words[2], words[3]};
  }
}

thread_local int g_tune_mode = 0;  // profiling only (ao_int4_set_tuning): 95-99 small-M A/B builds, 600-699 batched kernel (parts, ablation / trace builds)

template <int G, int MAXM, int DEPTH = 4>
int launch_mm(const uint16_t* x, const int32_t* qdata, const uint16_t* sz, uint16_t* y, int64_t M,
              int64_t N, int64_t K, hipStream_t stream, bool straight = false) {
  constexpr int ROWSTRIDE = (MAXM <= 4) ? 256 : 272;
  constexpr int SLAB = (MAXM + 1) * ROWSTRIDE;
  const int kblocks = (int)(K >> 7);
  const int64_t ntiles = N >> 4;
  const int64_t mslabs = (M + 15) / 16;
  // waves per workgroup: 8 once a wave still gets >= 2 weight blocks (measured at M = 1 on the Llama-3-8B shapes:
  // 4 waves 800, 8 waves 868, 16 waves 705 tok/s), else 4 (>= 256 threads for the epilogue)
  int wpb = (kblocks >= 16) ? 8 : 4;
  if (ntiles * mslabs >= 2048 && wpb > 4 && MAXM > 1) wpb /= 2;
  if (g_tune_wpb >= 4 && g_tune_wpb <= 16) wpb = g_tune_wpb;
  if (straight && kblocks % DEPTH == 0 && kblocks / DEPTH >= 4 && kblocks / DEPTH <= 16) wpb = kblocks / DEPTH;  // straight-line form
  if (MAXM > 4 && wpb > 8) wpb = 8;  // 16-row variant is built for <= 512 threads
  if (wpb > kblocks) wpb = kblocks < 4 ? 4 : kblocks;
  const size_t smem = (size_t)wpb * (SLAB + 1024);
  dim3 grid((unsigned)ntiles, (unsigned)mslabs), block(wpb * 64);
  auto go = [&](auto kern) {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(int4_mm_kernel)")) return rc;
    ao::launch(kern, grid, block, smem, stream, x, reinterpret_cast<const u32x4*>(qdata), reinterpret_cast<const uint32_t*>(sz), y,
               (int)M, (int)N, (int)K, (float*)nullptr, (unsigned*)nullptr);
    return (int)AO_OK;
  };
  if constexpr (MAXM == 1 && DEPTH != 4 && (DEPTH == 7 || DEPTH == 14 || DEPTH == 2 || DEPTH == 8 || DEPTH == 9)) {
    AO_REQUIRE(straight && kblocks == wpb * DEPTH && wpb <= 16, "int4_mm: the %d-deep form is straight-line only", DEPTH);
    if (int rc = go(int4_mm_kernel<G, MAXM, DEPTH, true>)) return rc;
  } else if constexpr (MAXM == 1 && DEPTH == 4) {
    if (straight && kblocks == wpb * DEPTH) { if (int rc = go(int4_mm_kernel<G, MAXM, DEPTH, true>)) return rc; }
    else if (int rc = go(int4_mm_kernel<G, MAXM, DEPTH, false>)) return rc;
  } else {
    if (int rc = go(int4_mm_kernel<G, MAXM, DEPTH, false>)) return rc;
  }
  AO_LAUNCH_CHECK("int4_mm_kernel launch");
  return AO_OK;
}

template <int G>
int dispatch_mm(const uint16_t* x, const int32_t* qdata, const uint16_t* sz, uint16_t* y, int64_t M,
                int64_t N, int64_t K, hipStream_t stream) {
  // M <= 16: one workgroup per 16-wide n-tile, waves split K (int4_mm_kernel, built for 1, 4 or 16 rows).  In the hipGraph
  // bench the single-row build beat both purpose-built decode kernels this round tried (a persistent balanced streaming kernel
  // with hand-counted LDS-DMA rings and a per-tile kernel with workgroup-shared x): 868 vs 732 tok/s.
  // At 4 < M <= 16 the per-tile kernel re-reads x for every 16 output columns (4 KiB of x per 1 KiB of weights at 16 rows): fine
  // for the narrow projections, 1.5x the M = 1 time on wide ones -- weights of >= 1024 n-tiles (merged gate_up_proj, lm_head) take
  // the batched kernel below with 16-row slabs, where 4 n-tiles share one staged x tile (gate_up_proj at M = 16: 32 -> 21.8 us).
  // (At M <= 4 the same switch measured 16.2 vs 17.4 us on one box and 19.7 vs 14.2 us on another: not taken.)
  // Modes 93-99: A/B builds for profiling (94 never / 93 always the batched kernel on wide weights at M <= 16, 97 the ring kernel at M = 1
  // whatever K, 98 the 4-row build at M = 1, 99 the 16-row build at any M).
  const bool wide = (N >> 4) >= 1024 && g_tune_mode != 94;
  if (M == 1 && g_tune_mode == 98) return launch_mm<G, 4>(x, qdata, sz, y, M, N, K, stream);
  if (g_tune_mode == 99) return launch_mm<G, 16>(x, qdata, sz, y, M, N, K, stream);
  // Round 6 (profiles/int4_forms_r06.jsonl, int4_forms_parts_r06.jsonl: 16 shapes of four models x M = 5 .. 16 x 9 forms, cold): the 16-row per-tile
  // build (9 .. 16 rows) is 1.3 - 1.6 x the 8-row build's time, and from 288 n-tiles on the batched kernel's 16-row slabs with K parts pass it --
  // gate 14336 x 4096 at M = 12 / 16 19.0 / 19.4 -> 14.1 us, 15360 x 5120 25.1 / 26.5 -> 16.6 / 16.8, 5120 x 13824 25.7 -> 17.4, 10240 x 8192
  // 24.0 / 25.3 -> 17.5 / 17.9, qkv 6144 x 4096 10.0 / 10.2 -> 9.5; at 224 - 256 n-tiles (o 4096^2: 6.2 against 8.5) and up to 8 rows the per-tile
  // kernel stays ahead.
  const bool tall = M > 8 && (N >> 4) >= 288 && g_tune_mode != 94;
  if (M <= 16 && g_tune_mode < 600 && !tall && !(wide && (M > 4 || g_tune_mode == 93))) {
    if (M == 1 && g_tune_mode != 97) {
      // round 3: when the weight's K divides into (waves <= 16) x (2 | 4 | 7 | 14 blocks), every block of the tile is requested in the
      // prologue and the kernel is straight-line code (58 - 64 VGPRs: four 8-wave workgroups per CU, so gate / up's 896 tiles are
      // resident at once; down_proj as 16 waves x 7 blocks has its whole weight in flight).  Llama-3-8B, same run, five shapes /
      // merged: 765 -> 818 / 846 -> 906 tok/s (profiles/int4_modes_r03.jsonl); mode 97 = the ring kernel everywhere
      const int64_t kbl = K >> 7;
      if (kbl % 4 == 0 && kbl / 4 >= 4 && kbl / 4 <= 16) return launch_mm<G, 1, 4>(x, qdata, sz, y, M, N, K, stream, true);
      if (kbl % 7 == 0 && kbl / 7 >= 4 && kbl / 7 <= 16) return launch_mm<G, 1, 7>(x, qdata, sz, y, M, N, K, stream, true);
      if (kbl % 14 == 0 && kbl / 14 >= 4 && kbl / 14 <= 16) return launch_mm<G, 1, 14>(x, qdata, sz, y, M, N, K, stream, true);
      if (kbl % 2 == 0 && kbl / 2 >= 4 && kbl / 2 <= 16) return launch_mm<G, 1, 2>(x, qdata, sz, y, M, N, K, stream, true);
      // round 6: K of other models that none of the four factors -- 8 blocks (K = 10240 / 12288 / 16384: 10 / 12 / 16 waves) and 9 (K = 13824,
      // Llama-2-13B's down_proj: 12 waves) -- had fallen to the ring kernel (mode 97 = that, for A/B: profiles/int4_depth_other_r06.jsonl)
      // Ahead on weights of up to 768 n-tiles (4096 x 12288 10.3 -> 9.1 us, 5120 x 13824 17.1 -> 16.0, 12288^2 22.8 -> 21.8, 6144 x 16384 19.5 -> 19.0),
      // behind on wider ones (13824 x 10240 23.7 -> 27.2, 14336 x 16384 35.2 -> 37.7: 10 - 16 waves per workgroup x 864+ workgroups), which keep the ring.
      if ((N >> 4) <= 768) {
        if (kbl % 8 == 0 && kbl / 8 >= 4 && kbl / 8 <= 16) return launch_mm<G, 1, 8>(x, qdata, sz, y, M, N, K, stream, true);
        if (kbl % 9 == 0 && kbl / 9 >= 4 && kbl / 9 <= 16) return launch_mm<G, 1, 9>(x, qdata, sz, y, M, N, K, stream, true);
      }
    }
    if (M == 1) return launch_mm<G, 1>(x, qdata, sz, y, M, N, K, stream);
    if (M <= 4) return launch_mm<G, 4>(x, qdata, sz, y, M, N, K, stream);
    // round 4: an 8-row build between the 4- and the 16-row one (half the x requests, LDS staging and slab of the 16-row build for batches
    // of 5 .. 8; mode 88 = the 16-row build as before, for A/B)
    if (M <= 8 && g_tune_mode != 88) return launch_mm<G, 8>(x, qdata, sz, y, M, N, K, stream);
    return launch_mm<G, 16>(x, qdata, sz, y, M, N, K, stream);
  }
  // 4 < M: int4_mm_rb_kernel on slabs of 16 / 32 / 64 / 128 rows (MT m-tiles: only the rows that exist are staged, read and
  // multiplied).  128-column tiles (8 waves) when that still gives ~a workgroup per CU; otherwise 64-column tiles (4 waves) cut
  // along K into at most 8 parts of >= 8 k-blocks so that the grid fills the chip (Llama-3-8B projections at M = 128: o_proj
  // 39.8 -> 14.6 us, down_proj 129 -> 29.6 us).  Modes 600 + 10 a + S (128-row slabs; wpb 8 / 4 waves, S parts, a = ablation /
  // trace build) and 700 + 10 log2(MT) + S: tuning / profiling.
  const int64_t kblocks = K >> 7;
  int mt = (M <= 16) ? 1 : (M <= 32) ? 2 : (M <= 64) ? 4 : 8;
  int forced_split = 0, waves = 0;
  if (g_tune_mode >= 600 && g_tune_mode < 700) {
    const int abl = (g_tune_mode - 600) / 10;
    mt = 8;
    waves = (g_tune_wpb == 4) ? 4 : 8;
    forced_split = std::max(1, g_tune_mode % 10);
    if constexpr (G == 128) {
      const int64_t base = ((N + 127) / 128) * ((M + 127) / 128);
      const int sp = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)forced_split, kblocks, kSplitMaxTiles / std::max<int64_t>(base, 1)}));
#ifdef AO_LAB  // 1 - 4: parts of the k-block removed (wrong results): laboratory library only
      if (abl == 1) return launch_mm_rb<G, 8, 1, 8, 1>(x, qdata, sz, y, M, N, K, sp, stream);
      if (abl == 2) return launch_mm_rb<G, 8, 1, 8, 2>(x, qdata, sz, y, M, N, K, sp, stream);
      if (abl == 3) return launch_mm_rb<G, 8, 1, 8, 3>(x, qdata, sz, y, M, N, K, sp, stream);
      if (abl == 4) return launch_mm_rb<G, 8, 1, 8, 4>(x, qdata, sz, y, M, N, K, sp, stream);
#endif
      if (abl == 5) return launch_mm_rb<G, 8, 1, 8, 5>(x, qdata, sz, y, M, N, K, sp, stream);
      if (abl == 6) return launch_mm_rb<G, 4, 2>(x, qdata, sz, y, M, N, K, sp, stream);
    }
  } else if (g_tune_mode >= 900 && g_tune_mode < 910 && M > 64) {
    // profiling: the producer-wave form (4 consumers + 4 DMA producers, 128-row slabs), 90S = S K-parts (900: as the product picks)
    const int64_t base9 = ((N + 63) / 64) * ((M + 127) / 128);
    const int64_t fit9 = (int64_t)kSplitMaxTiles * 128 * 128 / (base9 * 64 * 128);
    const int sp = (g_tune_mode == 900) ? (int)std::max<int64_t>(1, std::min<int64_t>({256 / base9, fit9, 8, kblocks / 8}))
                                        : (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(g_tune_mode - 900), kblocks, fit9}));
    return launch_mm_rb<G, 4, 1, 8, 0, true>(x, qdata, sz, y, M, N, K, sp, stream);
#ifdef AO_LAB
  } else if (g_tune_mode == 910 && M > 64) {
    // profiling: the same without the scale / zero DMAs (wrong numbers): what the dword LDS-DMAs cost
    const int64_t base9 = ((N + 63) / 64) * ((M + 127) / 128);
    const int64_t fit9 = (int64_t)kSplitMaxTiles * 128 * 128 / (base9 * 64 * 128);
    return launch_mm_rb<G, 4, 1, 8, 6, true>(x, qdata, sz, y, M, N, K, (int)std::max<int64_t>(1, std::min<int64_t>({256 / base9, fit9, 8, kblocks / 8})), stream);
#endif
  } else if (g_tune_mode >= 800 && g_tune_mode < 840 && M > 16) {
    // profiling (round 3): two n-tiles per wave -- every A fragment read from LDS feeds two MFMAs.  80S / 81S: 64-row slabs x 128
    // columns (4 waves, fused / with DMA-producer waves), 82S / 83S: 128-row slabs x 128 columns; S = K parts (0: fill ~256 workgroups)
    const int kind = (g_tune_mode - 800) / 10, s_req = g_tune_mode % 10;
    const int rows = (kind < 2) ? 64 : 128;
    const int64_t base8 = ((N + 127) / 128) * ((M + rows - 1) / rows);
    const int64_t fit8 = (int64_t)kSplitMaxTiles * 128 * 128 / (base8 * 128 * rows);
    const int sp = (int)std::max<int64_t>(1, s_req == 0 ? std::min<int64_t>({256 / base8, fit8, 8, kblocks / 8}) : std::min<int64_t>({(int64_t)s_req, kblocks, fit8}));
    if (kind == 0) return launch_mm_rb<G, 4, 2, 4>(x, qdata, sz, y, M, N, K, sp, stream);
    if (kind == 1) return launch_mm_rb<G, 4, 2, 4, 0, true>(x, qdata, sz, y, M, N, K, sp, stream);
    if (kind == 2) return launch_mm_rb<G, 4, 2, 8>(x, qdata, sz, y, M, N, K, sp, stream);
    return launch_mm_rb<G, 4, 2, 8, 0, true>(x, qdata, sz, y, M, N, K, sp, stream);
  } else if (g_tune_mode >= 920 && g_tune_mode < 940 && M > 64) {
    // round 5: the 128 x 128 / 32 x 32 x 16 kernel.  92S fused, 93S with DMA-producer waves; S = K parts (0: fill ~256 workgroups)
    const int s_req = g_tune_mode % 10;
    const int64_t base8 = ((N + 127) / 128) * ((M + 127) / 128);
    const int64_t fit8 = (int64_t)kSplitMaxTiles / base8;
    const int sp = (int)std::max<int64_t>(1, s_req == 0 ? std::min<int64_t>({256 / base8, fit8, 8, kblocks / 4}) : std::min<int64_t>({(int64_t)s_req, kblocks, fit8}));
    if (g_tune_mode < 930) return launch_mm_w32<G, false>(x, qdata, sz, y, M, N, K, sp, stream);
    return launch_mm_w32<G, true>(x, qdata, sz, y, M, N, K, sp, stream);
  } else if (g_tune_mode >= 950 && g_tune_mode < 960 && M > 64 && G >= 128) {
    // round 6: 64-column wave tiles (128 x 256 workgroup tiles, every A fragment feeds two MFMAs), producer form; 95S: S K parts (0: fill ~256 workgroups).
    // (The form in which the consumers fetch for themselves -- 4 waves, 424 registers, no spills -- was 5 - 25 % behind this one in every cell of
    // profiles/int4_w64_ab_r06.jsonl (modes 96S there) and left the tree.)
    if constexpr (G >= 128) {
      const int s_req = g_tune_mode % 10;
      const int64_t base8 = ((N + 255) / 256) * ((M + 127) / 128);
      const int64_t fit8 = (int64_t)kSplitMaxTiles / (2 * base8);
      const int sp = (int)std::max<int64_t>(1, s_req == 0 ? std::min<int64_t>({256 / base8, fit8, 8, kblocks / 4}) : std::min<int64_t>({(int64_t)s_req, kblocks, fit8}));
      return launch_mm_w32<G, true, 0, 2>(x, qdata, sz, y, M, N, K, sp, stream);
    }
  } else if (g_tune_mode >= 940 && g_tune_mode < 950 && M > 64) {
    // profiling: 945 = the producer form with s_memtime stamps; 941 / 942 / 943 (laboratory library only: wrong results) = without the
    // 32 x 32 x 16 MFMAs / the dequant / the A-fragment reads.  One K part.
    if constexpr (G == 128) {
#ifdef AO_LAB
      if (g_tune_mode == 941) return launch_mm_w32<G, true, 1>(x, qdata, sz, y, M, N, K, 1, stream);
      if (g_tune_mode == 942) return launch_mm_w32<G, true, 2>(x, qdata, sz, y, M, N, K, 1, stream);
      if (g_tune_mode == 943) return launch_mm_w32<G, true, 3>(x, qdata, sz, y, M, N, K, 1, stream);
#endif
      if (g_tune_mode == 945) return launch_mm_w32<G, true, 5>(x, qdata, sz, y, M, N, K, 1, stream);
    }
    return launch_mm_w32<G, true>(x, qdata, sz, y, M, N, K, 1, stream);
  } else if (g_tune_mode >= 700 && g_tune_mode < 800) {
    mt = 1 << std::min(3, (g_tune_mode - 700) / 10);
    waves = (g_tune_wpb == 8 && mt >= 2) ? 8 : 4;
    forced_split = std::max(1, g_tune_mode % 10);
  }
  // Round 5: the 128 x 128 / 32 x 32 x 16 kernel with DMA-producer waves (int4_mm_w32_kernel) from 129 rows on, and on wide weights
  // (>= 64 column tiles) from 65 rows.  profiles/int4_w32_ab_r05.jsonl, cold, us (this dispatch before -> after): M = 256 qkv 32.8 -> 26.9,
  // o 21.6 -> 21.5, gate 49.2 -> 37.9, down 56.1 -> 44.2; M = 2048 gate 320 -> 281, down 301 -> 267; M = 128 gate 29.0 -> 28.2 (qkv, o, down
  // stay: 48 / 32 column tiles need 5 - 8 K parts of 64 KiB partial tiles to fill the chip, 20.4 / 15.0 / 31.7 vs 22.0 / 21.7 / 34.1).
  // Mode 911: never (the round-4 dispatch, for A/B).
  if (g_tune_mode != 911 && (g_tune_mode < 600 || g_tune_mode == 912) && forced_split == 0 && waves == 0 && M > 64 && (M > 128 || (N + 127) / 128 >= 64)) {
    // Round 6: 64-column wave tiles (128 x 256 workgroup tiles) where one K part of them fills whole rounds of the chip -- the tile is twice as
    // large, so the last round has to be >= 7/8 full -- from 512 rows, group sizes >= 128 (int4_mm_w32_tiles64).  profiles/int4_w64_ab_r06.jsonl,
    // cold, us: M = 512 gate 70.7 -> 64.1; M = 2048 o 79.7 -> 72.8, down 262 -> 238, gate 278 -> 266; qkv at M = 2048 (384 tiles = 1.5 rounds)
    // 117 -> 121 and every shape below 200 tiles (K parts of a 128 KiB partial tile) 1.3 - 4 x slower: those keep the 128 x 128 tile.  Mode 912: never.
    if constexpr (G >= 128) {
      if (g_tune_mode != 912 && int4_mm_w32_tiles64(M, N)) return launch_mm_w32<G, true, 0, 2>(x, qdata, sz, y, M, N, K, 1, stream);
    }
    const int64_t base8 = ((N + 127) / 128) * ((M + 127) / 128);
    const int64_t fit8 = (int64_t)kSplitMaxTiles / base8;
    const int sp = (int)std::max<int64_t>(1, std::min<int64_t>({256 / base8, fit8, 8, kblocks / 4}));
    return launch_mm_w32<G, true>(x, qdata, sz, y, M, N, K, sp, stream);
  }
  const int64_t slabs = (M + 16 * mt - 1) / (16 * mt);
  // (32-row slabs: 4 waves x K parts beat the 8-wave tile that cuts nothing -- 28672 x 8192 at M = 24 / 32: 53 -> 43.5 us)
  if (waves == 0) waves = (mt >= 4 && ((N + 127) / 128) * slabs >= 190) ? 8 : 4;
  const int bn = waves * 16;
  const int64_t base = ((N + bn - 1) / bn) * slabs;
  const int64_t fit = (int64_t)kSplitMaxTiles * 128 * 128 / (base * bn * 16 * mt);
  int split = 1;
  if (forced_split > 0) split = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)forced_split, kblocks, fit}));
  // (16-row slabs are light -- 27 KiB of LDS, 4 waves -- so several workgroups share a CU: aim for ~1000 of them)
  // (round 6, the same files: with 16-row slabs at least 4 parts up to 512 column tiles -- 18944 x 3584: 3 -> 4 parts 15.9 -> 15.0 us, 28672 x 8192:
  // 2 -> 4 41.4 -> 36.9 -- and parts down to 4 k-blocks: 5120 x 5120 5 -> 8 parts 10.7 -> 9.6)
  // 32-row slabs the same; 64-row slabs aim for ~512 workgroups (was 256) -- profiles/int4_forms_mt2_r06.jsonl, int4_forms_mt4_r06.jsonl, M = 24 / 32 and
  // 48 / 64, us before -> with this rule's count: qkv 6144 x 4096 13.1 -> 10.5 and 15.4 -> 13.5, gate 14336 x 4096 20.4 -> 16.2, down 4096 x 14336
  // 20.0 -> 16.5 and 23.8 -> 20.3, 10240 x 8192 35.3 -> 20.2 and 39.6 -> ~31, 8192 x 28672 65.1 -> 43.4 and 77.5 -> 59.1; o 4096^2 level.
  else if (waves == 4 && mt <= 2)
    split = (int)std::max<int64_t>(1, std::min<int64_t>({std::max<int64_t>(base <= 512 ? 4 : 1, 1024 / base), fit, 8, kblocks / 4}));
  else if (waves == 4 && mt == 4) split = (int)std::max<int64_t>(1, std::min<int64_t>({512 / base, fit, 8, kblocks / 8}));
  else if (waves == 4) split = (int)std::max<int64_t>(1, std::min<int64_t>({256 / base, fit, 8, kblocks / 8}));
  if (waves == 8) {
    if (mt == 8) return launch_mm_rb<G, 8, 1, 8>(x, qdata, sz, y, M, N, K, split, stream);
    if (mt == 4) return launch_mm_rb<G, 8, 1, 4>(x, qdata, sz, y, M, N, K, split, stream);
    return launch_mm_rb<G, 8, 1, 2>(x, qdata, sz, y, M, N, K, split, stream);
  }
  if (mt == 8) {
    // one 128-row slab (65 .. 128 rows: the "bs = 128" half of the BASELINE metric): the form with DMA-producer waves -- Llama-3-8B
    // at bs = 128 32.3k -> 35.4k tok/s (gate_proj 28.2 -> 25.5 us, same bits); with two or more slabs the two forms measure the
    // same (bs = 256) or the fused form wins (bs = 2048: 57k vs 50k tok/s).  Mode 910: never.
    if (slabs == 1 && g_tune_mode != 910) return launch_mm_rb<G, 4, 1, 8, 0, true>(x, qdata, sz, y, M, N, K, split, stream);
    return launch_mm_rb<G, 4, 1, 8>(x, qdata, sz, y, M, N, K, split, stream);
  }
  if (mt == 4) return launch_mm_rb<G, 4, 1, 4>(x, qdata, sz, y, M, N, K, split, stream);
  if (mt == 2) return launch_mm_rb<G, 4, 1, 2>(x, qdata, sz, y, M, N, K, split, stream);
  return launch_mm_rb<G, 4, 1, 1>(x, qdata, sz, y, M, N, K, split, stream);
}

int check_int4_shape(const char* fn, int64_t N, int64_t K, int group_size) {
  AO_REQUIRE(N > 0 && K > 0, "%s: N and K must be positive, got N=%lld K=%lld", fn, (long long)N, (long long)K);
  AO_REQUIRE(N % 16 == 0, "%s: N=%lld must be a multiple of 16 (gfx950 n-tile)", fn, (long long)N);
  AO_REQUIRE(K % 128 == 0, "%s: K=%lld must be a multiple of inner_k_tiles*16 = 128", fn, (long long)K);
  if (group_size != 0) {
    AO_REQUIRE(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256,
               "%s: qGroupSize must be one of 32, 64, 128, 256, got %d", fn, group_size);
    AO_REQUIRE(K % group_size == 0, "%s: K=%lld not divisible by qGroupSize=%d", fn, (long long)K, group_size);
  }
  AO_REQUIRE(N < (1ll << 31) && K < (1ll << 31), "%s: N, K must fit int32", fn);
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" const char* ao_int4_mm_kernel_name(int64_t M, int64_t N, int64_t K, int group_size) {
  (void)K;
  if (M > 64 && (M > 128 || (N + 127) / 128 >= 64))  // round 5: 128 x 128 tiles, 32 x 32 x 16 MFMAs; round 6: 128 x 256 where they fill whole rounds
    return (group_size >= 128 && int4_mm_w32_tiles64(M, N)) ? "int4_mm_w32_kernel<128x256>" : "int4_mm_w32_kernel";
  if (M > 16 || (M > 4 && (N >> 4) >= 1024) || (M > 8 && (N >> 4) >= 288)) return "int4_mm_rb_kernel";  // (round 6: 9 .. 16 rows from 288 n-tiles)
  return "int4_mm_kernel";
}

extern "C" int ao_int4_set_trace(unsigned long long* trace_dev) {
  g_mm_trace = trace_dev;
  fp8_rowwise_rb_set_trace(trace_dev);
  return AO_OK;
}

namespace ao { void fp8_int4_set_mt(int mt, bool nt1); }  // fp8_int4_kernels.hip
extern "C" int ao_int4_set_tuning(int waves_per_block, int mode) {
  g_tune_wpb = waves_per_block;
  g_tune_mode = mode;
  // fp8-act x int4: 961 / 962 / 964 m-tiles per workgroup forced; 970 / 972 / 974: one n-tile per workgroup (0 / 2 / 4 m-tiles: 0 = by M)
  ao::fp8_int4_set_mt((mode == 961 || mode == 962 || mode == 964) ? mode - 960 : (mode == 972 || mode == 974) ? mode - 970 : 0, mode >= 970 && mode <= 974);
  return AO_OK;
}

extern "C" int ao_int4_convert_weight_to_int4pack(const uint8_t* w_u8, int32_t* qdata, int64_t N,
                                                  int64_t K, int inner_k_tiles, void* stream) {
  AO_REQUIRE_PTR(w_u8);
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE(inner_k_tiles == 8, "ao_int4_convert_weight_to_int4pack: innerKTiles must be 8 (torchao fixes it), got %d", inner_k_tiles);
  if (int rc = check_int4_shape(__func__, N, K, 0)) return rc;
  const int64_t words = N * K / 8;
  const int threads = 256;
  ao::launch(int4_pack_kernel, dim3((unsigned)((words + threads - 1) / threads)), dim3(threads), 0,
                     (hipStream_t)stream, w_u8, reinterpret_cast<uint32_t*>(qdata), N, K, words);
  AO_LAUNCH_CHECK("int4_pack_kernel launch");
  return AO_OK;
}

extern "C" int ao_int4_unpack_int4pack(const int32_t* qdata, uint8_t* w_u8, int64_t N, int64_t K,
                                       int inner_k_tiles, void* stream) {
  AO_REQUIRE_PTR(w_u8);
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE(inner_k_tiles == 8, "ao_int4_unpack_int4pack: innerKTiles must be 8, got %d", inner_k_tiles);
  if (int rc = check_int4_shape(__func__, N, K, 0)) return rc;
  const int64_t words = N * K / 8;
  const int threads = 256;
  ao::launch(int4_unpack_kernel, dim3((unsigned)((words + threads - 1) / threads)), dim3(threads), 0,
                     (hipStream_t)stream, reinterpret_cast<const uint32_t*>(qdata), w_u8, N, K, words);
  AO_LAUNCH_CHECK("int4_unpack_kernel launch");
  return AO_OK;
}

extern "C" int ao_int4_weight_int4pack_mm(const uint16_t* x, const int32_t* qdata,
                                          const uint16_t* scale_and_zero, uint16_t* y, int64_t M,
                                          int64_t N, int64_t K, int group_size, void* stream) {
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE_PTR(scale_and_zero);
  if (int rc = check_int4_shape(__func__, N, K, group_size)) return rc;
  AO_REQUIRE(M >= 0 && M < (1ll << 31), "ao_int4_weight_int4pack_mm: bad M=%lld", (long long)M);
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(x);
  AO_REQUIRE_PTR(y);
  hipStream_t s = (hipStream_t)stream;
  // The batched kernel addresses x rows with 32-bit byte offsets and grid.y is 16-bit: very tall activations go in row
  // chunks (a multiple of 128 rows, < 4 GiB of x and <= 65535 slabs each), one launch per chunk.
  const int64_t by_bytes = ((1ll << 32) - 1) / (2 * K) / 128 * 128;
  const int64_t chunk = std::max<int64_t>(128, std::min<int64_t>(by_bytes, 65535ll * 16));
  for (int64_t m0 = 0; m0 < M; m0 += chunk) {
    const int64_t rows = std::min(chunk, M - m0);
    const uint16_t* xc = x + m0 * K;
    uint16_t* yc = y + m0 * N;
    int rc;
    switch (group_size) {
      case 32: rc = dispatch_mm<32>(xc, qdata, scale_and_zero, yc, rows, N, K, s); break;
      case 64: rc = dispatch_mm<64>(xc, qdata, scale_and_zero, yc, rows, N, K, s); break;
      case 128: rc = dispatch_mm<128>(xc, qdata, scale_and_zero, yc, rows, N, K, s); break;
      default: rc = dispatch_mm<256>(xc, qdata, scale_and_zero, yc, rows, N, K, s); break;
    }
    if (rc != AO_OK) return rc;
  }
  return AO_OK;
}

extern "C" int ao_int4_dequantize(const int32_t* qdata, const uint16_t* scale_and_zero,
                                  uint16_t* w_bf16, int64_t N, int64_t K, int group_size,
                                  void* stream) {
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE_PTR(scale_and_zero);
  AO_REQUIRE_PTR(w_bf16);
  if (int rc = check_int4_shape(__func__, N, K, group_size)) return rc;
  const int64_t words = N * K / 8;
  const int threads = 256;
  dim3 grid((unsigned)((words + threads - 1) / threads)), block(threads);
  const uint32_t* q = reinterpret_cast<const uint32_t*>(qdata);
  const uint32_t* sz = reinterpret_cast<const uint32_t*>(scale_and_zero);
  hipStream_t s = (hipStream_t)stream;
  switch (group_size) {
    case 32: ao::launch(int4_dequant_kernel<32>, grid, block, 0, s, q, sz, w_bf16, N, K, words); break;
    case 64: ao::launch(int4_dequant_kernel<64>, grid, block, 0, s, q, sz, w_bf16, N, K, words); break;
    case 128: ao::launch(int4_dequant_kernel<128>, grid, block, 0, s, q, sz, w_bf16, N, K, words); break;
    default: ao::launch(int4_dequant_kernel<256>, grid, block, 0, s, q, sz, w_bf16, N, K, words); break;
  }
  AO_LAUNCH_CHECK("int4_dequant_kernel launch");
  return AO_OK;
}

extern "C" int ao_int4_quantize_tinygemm(const uint16_t* w, int32_t* qdata, uint16_t* scale_and_zero,
                                         int64_t N, int64_t K, int group_size, void* stream) {
  AO_REQUIRE_PTR(w);
  AO_REQUIRE_PTR(qdata);
  AO_REQUIRE_PTR(scale_and_zero);
  if (int rc = check_int4_shape(__func__, N, K, group_size)) return rc;
  hipStream_t s = (hipStream_t)stream;
  u32x4* q = reinterpret_cast<u32x4*>(qdata);
  uint32_t* sz = reinterpret_cast<uint32_t*>(scale_and_zero);
  const int64_t kb_per = group_size > 128 ? group_size / 128 : 1;
  const int64_t spans = (N / 16) * ((K / 128) / kb_per);
  AO_REQUIRE(spans < (1ll << 31), "ao_int4_quantize_tinygemm: tensor too large for one launch");
  dim3 grid((unsigned)spans), block(64);
  switch (group_size) {
    case 32: ao::launch(int4_quantize_kernel<32>, grid, block, 0, s, w, q, sz, N, K); break;
    case 64: ao::launch(int4_quantize_kernel<64>, grid, block, 0, s, w, q, sz, N, K); break;
    case 128: ao::launch(int4_quantize_kernel<128>, grid, block, 0, s, w, q, sz, N, K); break;
    default: ao::launch(int4_quantize_kernel<256>, grid, block, 0, s, w, q, sz, N, K); break;
  }
  AO_LAUNCH_CHECK("int4_quantize_kernel launch");
  return AO_OK;
}
