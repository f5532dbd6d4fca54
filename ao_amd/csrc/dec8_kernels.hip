// 8-bit decode linears (M <= 16) on gfx950, round 4: the weight stream as FULL 128-byte lines in a register ring.
//
//   int8:  y = bf16(bf16(i32(xq . wq^T) * sx[m]) * sw[n] + bias[n])     Int8Tensor F.linear, int8_tensor.py:305-359 (+ :176-248 cast)
//   fp8 :  y = bf16((xq . wq^T) * sx[m] * sw[n] + bias[n])              Float8Tensor rowwise, float8/inference.py:104-123
//          (+ float8_tensor.py:167-253 cast)
//
// Bound: HBM -- 1 byte per weight, read once.  What the round-3 kernels (stream8_kernel / dyn8_kernel) left on the table, found by the
// disassembly scan at the end of that round: their k loop loaded ONE 2 KiB step, waited vmcnt(0), multiplied, looped; and every weight
// load instruction touched 16 different 128-byte lines for 16 bytes each (lane = (row n, 16-byte piece kq): the MFMA operand layout),
// so a 2 KiB step cost the texture path 128 line look-ups for 32 lines of data.
//
// Here: one workgroup per 16-row n-tile, its waves split K in contiguous runs of DEPTH steps (K = 128 * DEPTH * waves: straight-line
// code, no loop), and
//   * every wave requests its whole run in the prologue -- 2 x global_load_dwordx4 per step in which 8 consecutive lanes cover one
//     full line (lane l: row 8 i + (l >> 3), 16-byte chunk l & 7) -- DEPTH x 2 KiB per wave in flight in a static register ring,
//     the compiler counts the vmcnt(N) waits;
//   * the activation is cast (DYN: amax -> scale -> codes with quant_math.h's arithmetic, one row held in registers; its loads are
//     issued BEFORE the ring's so that they are not queued behind 128 KiB of weights) or copied ONCE per workgroup into LDS while the
//     weights are in flight;
//   * a step is turned into the MFMA's operand layout through a wave-private 2.25 KiB LDS slab (rows 144 bytes apart: the
//     ds_read_b128 of 16 rows hit 64 distinct banks): 2 ds_write_b128 + 2 ds_read_b128, no barrier (DS ops of a wave run in order);
//   * operands keep stream8_kernel's k assignment (lane group kq: k = 16 kq .. +15 and 64 + 16 kq .. +15), so a wave's accumulator
//     sees the same products in the same MFMA as before; rows >= M of the A operand alias row 0 (their outputs are never stored);
//   * one barrier for the cross-wave reduction (wave order: reproducible), the reference's epilogues in fp32, bf16 out.
#include "common.h"
#include "quant_math.h"

#include <algorithm>
#include <type_traits>

namespace ao {
thread_local int g_dec8_mode = 0;  // ao_gemm8_set_variant 200 + d: force ring depth d; 290: half-line loads (no LDS transposition); 291 / 292: 8-row tiles never / always; 299: never this kernel
namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct Dec8Args {
  const void* x;           // DYN: bf16 [M][K];  else codes [M][K]
  const float* row_scale;  // !DYN: fp32 [M]
  const uint8_t* b;        // [N][K] int8 / e4m3
  const float* col_scale;  // [N]
  const uint16_t* bias;    // [N] bf16 or null
  uint16_t* out;           // [M][N] bf16
  int M, N, K;
};

// workgroup barrier that orders LDS traffic only: __syncthreads() carries a workgroup-scope release fence, which on gfx9 means
// s_waitcnt vmcnt(0) -- it would drain the weight ring that is in flight across every barrier of this kernel
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kSlabStride = 144;            // bytes between the 16 rows of a wave's transposition slab
constexpr int kSlab = 16 * kSlabStride;     // 2304 B
constexpr int kXV = 4;                      // 16-byte activation vectors a thread may hold while the ring is in flight

// XFAST: a wave holds the activation slice of its own k-run in registers (DYN: M == 1; pre-quantized: M x DEPTH <= 32) -- those loads
// go out first, the ring behind them, cast / copy run under the weights' flight and stay wave-private.  Otherwise the workgroup-wide
// cast / copy loops run first and the ring is requested after them.
// LOOP: any K % 128 == 0 -- a wave's run is ceil / floor (K / 128 / waves) steps, walked with the DEPTH-deep ring refilled slot by slot
// (steady state branch-free, so the compiler still counts its vmcnt waits; the drain has uniform branches).  K of the popular models
// that do not factor into <= 16 waves x {8, 7, 4, 2, 1} steps take it: Llama-2-7B's 11008 (86 steps), Llama-3-70B's unsharded 28672.
// ROWS8 (round 6): the tile is 8 weight rows, not 16 -- one full-line load per step; columns 8 .. 15 of the MFMA repeat columns 0 .. 7 and are
// not stored.  For weights with so few rows that N / 16 workgroups leave most CUs without work (the 70B / TP8 qkv shard: 80 workgroups).
template <bool INT8, bool DYN, int DEPTH, bool XFAST, bool HALF, bool LOOP = false, bool ROWS8 = false>
__global__ __launch_bounds__(1024) void dec8_kernel(Dec8Args p) {
  static_assert(!(LOOP && XFAST), "the loop form casts / copies the activation workgroup-wide");
  static_assert(!(ROWS8 && (HALF || LOOP)), "8-row tiles: the straight-line full-line form only");
  constexpr int TR = ROWS8 ? 8 : 16;  // weight rows per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x, nwaves = nthreads >> 6;
  const int stride = p.K + 16;  // bytes between rows of codes: (K + 16) / 4 = 4 (mod 64) banks for K % 256 == 0
  char* xq = smem;                                                              // [M][K + 16]
  char* slab = smem + ((p.M * stride + 15) & ~15) + wave * kSlab;               // [nwaves][16][144]
  float* red = reinterpret_cast<float*>(smem + ((p.M * stride + 15) & ~15) + nwaves * kSlab);  // [nwaves][256]
  float* wmax = red + nwaves * 256;                                             // [nwaves][16]
  float* rs = wmax + nwaves * 16;                                               // [16]

  const int ntile = blockIdx.x;
  const int ksteps = p.K >> 7;
  const int ks0 = LOOP ? (ksteps * wave) / nwaves : wave * DEPTH;  // this wave's first 128-k step
  const int ks1 = LOOP ? (ksteps * (wave + 1)) / nwaves : ks0 + DEPTH;
  const int kq = lane >> 4, nl = lane & 15;

  struct Stage {
    u32x4 b0, b1;
  };
  Stage st[DEPTH];
  // lane l: HALF -- row l & 15, pieces kq and 4 + kq of the step (the operand layout);  else row (l >> 3) of the tile's rows 0..7 (b0) /
  // 8..15 (b1), chunk l & 7 of the step's 128 bytes: full lines
  const uint8_t* brow = (HALF ? p.b + ((size_t)ntile * 16 + nl) * p.K + kq * 16 : p.b + ((size_t)ntile * TR + (lane >> 3)) * p.K + (lane & 7) * 16) +
                        (size_t)ks0 * 128;  // the wave's first step; steps are addressed relative to it (straight-line form: immediates)
  const size_t b1off = HALF ? (size_t)64 : (size_t)8 * p.K;
  auto issue_step = [&](Stage& s, int rel) {
    s.b0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(brow + rel * 128));
    if constexpr (!ROWS8) s.b1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(brow + b1off + rel * 128));
    __builtin_amdgcn_sched_barrier(0);  // request order = consumption order (VMEM returns in order)
  };
  auto issue_ring = [&]() {
    const int last = max(ks1 - ks0 - 1, 0);  // (LOOP: a wave with fewer than DEPTH steps re-reads its last one, unused)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue_step(st[d], LOOP ? min(d, last) : d);
    // nothing that waits for an earlier load (the activation's) may be scheduled above the ring's requests
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- 1. activation -> codes in LDS (once per workgroup), weights requested as early as the VMEM queue order allows
  if constexpr (DYN && XFAST) {
    // M == 1.  A wave multiplies only its own k-run, so it casts only that: DEPTH x 128 bf16 = DEPTH x 16 vectors of 8, held in
    // registers between amax and cast and written to the wave's own part of the code row -- the one cross-wave dependency left is the
    // row's amax (ONE LDS-only barrier; the codes need none: DS operations of a wave execute in order).
    constexpr int XW = (DEPTH * 16 + 63) / 64;  // vectors per lane
    const u32x4* xr = reinterpret_cast<const u32x4*>(p.x) + (size_t)ks0 * 16;
    u32x4 xv[XW];
#pragma unroll
    for (int i = 0; i < XW; ++i) xv[i] = xr[min(lane + i * 64, DEPTH * 16 - 1)];  // clamped, unconditional: straight-line vmcnt
    __builtin_amdgcn_sched_barrier(0);
    issue_ring();
    float m = 0.f;
    bool has_nan = false;
#pragma unroll
    for (int i = 0; i < XW; ++i) m = fmaxf(m, amax8(xv[i], has_nan));
    if (has_nan) m = INFINITY;  // (NaN rows are outside the contract, as in the stand-alone cast)
    m = wave_max(m);
    if (lane == 0) wmax[wave * 16] = m;
    lds_barrier();
    float mm = 0.f;
    for (int w = 0; w < nwaves; ++w) mm = fmaxf(mm, wmax[w * 16]);
    const float s = INT8 ? int8_row_scale(mm) : fp8_row_scale(mm);
    const float inv = 1.0f / s;
    if (tid == 0) rs[0] = s;
#pragma unroll
    for (int i = 0; i < XW; ++i) {
      const int idx = lane + i * 64;
      if (idx < DEPTH * 16) *reinterpret_cast<u32x2*>(xq + ks0 * 128 + idx * 8) = INT8 ? int8_quant8(xv[i], inv) : fp8_quant8(xv[i], s);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private codes: no barrier)
  } else if constexpr (DYN) {
    // dyn8_kernel's two passes over the L2-resident activation, then the ring
    const int nvec = p.K >> 3;
    const uint16_t* x = reinterpret_cast<const uint16_t*>(p.x);
    for (int r = 0; r < p.M; ++r) {
      const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)r * p.K);
      float m = 0.f;
      bool has_nan = false;
      for (int i = tid; i < nvec; i += nthreads) m = fmaxf(m, amax8(xr[i], has_nan));
      if (has_nan) m = INFINITY;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      if (lane == 0) wmax[wave * 16 + r] = m;
    }
    lds_barrier();
    if (tid < p.M) {
      float m = 0.f;
      for (int w = 0; w < nwaves; ++w) m = fmaxf(m, wmax[w * 16 + tid]);
      rs[tid] = INT8 ? int8_row_scale(m) : fp8_row_scale(m);
    }
    lds_barrier();
    for (int r = 0; r < p.M; ++r) {
      const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)r * p.K);
      const float s = rs[r];
      const float inv = 1.0f / s;
      for (int i = tid; i < nvec; i += nthreads)
        *reinterpret_cast<u32x2*>(xq + r * stride + i * 8) = INT8 ? int8_quant8(xr[i], inv) : fp8_quant8(xr[i], s);
    }
    issue_ring();
  } else if constexpr (XFAST) {
    // codes [M][K]: a wave copies the part of every row that ITS k-run multiplies (M x DEPTH x 8 vectors of 16 bytes, at most kXV
    // per lane) -- wave-private, so no barrier stands between the copy and the first MFMA
    const int nvec = p.M * DEPTH * 8;
    u32x4 xv[kXV];
    int off[kXV];
#pragma unroll
    for (int i = 0; i < kXV; ++i) {
      const int v = min(lane + i * 64, nvec - 1);
      const int r = v / (DEPTH * 8), c = v - r * (DEPTH * 8);
      off[i] = r * stride + ks0 * 128 + c * 16;
      xv[i] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.x) + (size_t)r * p.K + ks0 * 128 + c * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_ring();
#pragma unroll
    for (int i = 0; i < kXV; ++i)
      if (lane + i * 64 < nvec) *reinterpret_cast<u32x4*>(xq + off[i]) = xv[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    const int vpr = p.K >> 4, nvec = p.M * vpr;
    const u32x4* xr = reinterpret_cast<const u32x4*>(p.x);
    for (int idx = tid; idx < nvec; idx += nthreads) {
      const int r = idx / vpr, c = idx - r * vpr;
      *reinterpret_cast<u32x4*>(xq + r * stride + c * 16) = xr[idx];
    }
    issue_ring();
  }
  if constexpr (!XFAST) lds_barrier();  // (XFAST: every wave wrote the codes it reads itself)

  // ---- 2. one pass of the ring: transpose a step through the slab, multiply
  // rows >= M of the A operand alias row 0: they only reach output rows that are never stored
  const char* arow = xq + (nl < p.M ? nl : 0) * stride + kq * 16 + ks0 * 128;  // relative steps, like the weights
  char* wr0 = slab + (lane >> 3) * kSlabStride + (lane & 7) * 16;  // this lane's piece of rows 0..7; rows 8..15: + 8 rows
  const char* rd = slab + (ROWS8 ? (nl & 7) : nl) * kSlabStride + kq * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // int8: int32 bit patterns
  auto consume = [&](const Stage& sg, int rel) {
    u32x4 b0, b1;
    if constexpr (HALF) {
      b0 = sg.b0; b1 = sg.b1;
    } else {
      *reinterpret_cast<u32x4*>(wr0) = sg.b0;
      if constexpr (!ROWS8) *reinterpret_cast<u32x4*>(wr0 + 8 * kSlabStride) = sg.b1;
      b0 = *reinterpret_cast<const u32x4*>(rd);
      b1 = *reinterpret_cast<const u32x4*>(rd + 64);
    }
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(arow + rel * 128);
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(arow + rel * 128 + 64);
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
  };
  if constexpr (!LOOP) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) consume(st[d], d);
  } else {
    const int n = ks1 - ks0;
    int r = 0;
    for (; r + 2 * DEPTH <= n; r += DEPTH) {  // steady state: every consumed slot is refilled, no branch in the body
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        consume(st[d], r + d);
        issue_step(st[d], r + d + DEPTH);
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {  // drain: fewer than 2 DEPTH steps left
      if (r + d < n) {
        consume(st[d], r + d);
        if (r + d + DEPTH < n) issue_step(st[d], r + d + DEPTH);
      }
    }
    r += DEPTH;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (r + d < n) consume(st[d], r + d);
  }

  // ---- 3. split-K reduction across waves (wave order: reproducible), scales, store
  {
    float* r = red + (size_t)wave * 256 + (kq * 4) * 16 + nl;  // [row 16][col 16]
    r[0] = acc.x; r[16] = acc.y; r[32] = acc.z; r[48] = acc.w;
  }
  lds_barrier();
  for (int idx = tid; idx < p.M * 16; idx += nthreads) {  // (workgroups of 1 .. 3 waves have fewer threads than outputs)
    const int row = idx >> 4, col = idx & 15;
    if (ROWS8 && col >= 8) continue;
    const int gn = ntile * TR + col;
    const float sx = DYN ? rs[row] : p.row_scale[row];
    float v;
    if constexpr (INT8) {
      int isum = 0;
      for (int w = 0; w < nwaves; ++w) isum += __float_as_int(red[(size_t)w * 256 + idx]);
      // t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))   (int8_tensor.py:315-359)
      v = round_bf16((float)isum * sx) * p.col_scale[gn];
    } else {
      float sum = 0.f;
      for (int w = 0; w < nwaves; ++w) sum += red[(size_t)w * 256 + idx];
      v = sum * sx * p.col_scale[gn];
    }
    if (p.bias != nullptr) v += bf16_lo_to_f32(p.bias[gn]);
    p.out[(size_t)row * p.N + gn] = f32_to_bf16_bits(v);
  }
}

struct Dec8Shape {
  int waves, depth;
  bool loop;  // K does not factor: the ring is refilled in a loop (depth 4)
};

// K = 128 * depth * waves: the deepest ring of {8, 7, 4, 2, 1} that leaves at most 16 waves.  (Measured on the 70B / TP8 fp8 shards and the
// Llama-3-8B int8 shapes, profiles/dec8_forms_r04.txt: the deeper ring wins at every K, down to ONE wave x 8 steps for K = 1024 --
// fewer waves mean fewer partials to reduce and more workgroups per CU.)
bool dec8_shape(int64_t K, int forced_depth, Dec8Shape* out) {
  if (K % 128 != 0) return false;
  const int ksteps = (int)(K / 128);
  static const int depths[] = {8, 7, 4, 2, 1};
  for (int d : depths) {
    if (forced_depth != 0 && d != forced_depth) continue;
    if (ksteps % d != 0) continue;
    const int w = ksteps / d;
    if (w >= 1 && w <= 16) {
      *out = Dec8Shape{w, d, false};
      return true;
    }
  }
  if (forced_depth != 0) return false;
  // no such factorization (K = 11008: 86 steps, K = 28672: 224 steps): up to 16 waves walk uneven runs with a 4-deep ring in a loop
  *out = Dec8Shape{(int)std::min<int64_t>(16, std::max<int64_t>(1, ksteps / 4)), 4, true};
  return true;
}

size_t dec8_lds(int64_t M, int64_t K, int waves) {
  return (size_t)((M * (K + 16) + 15) & ~(int64_t)15) + (size_t)waves * kSlab + (size_t)(waves * 256 + waves * 16 + 16) * sizeof(float);
}

// 8-row tiles where 16-row tiles would leave more than half of the chip's CUs without a workgroup (N / 16 < 128) -- the straight-line
// form with more than one wave only (a one-wave workgroup streams 1 KiB per step as it is)
bool dec8_rows8(int64_t N, int waves) { return g_dec8_mode != 291 && (g_dec8_mode == 292 || N / 16 < 128) && N % 8 == 0 && waves >= 2; }

template <bool INT8, bool DYN, int DEPTH, bool XFAST>
int launch_dec8_h(const Dec8Args& p, int waves, bool half, hipStream_t stream) {
  const size_t smem = dec8_lds(p.M, p.K, waves);
  if (!half && dec8_rows8(p.N, waves)) {
    auto kern = dec8_kernel<INT8, DYN, DEPTH, XFAST, false, false, true>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(dec8_kernel)")) return rc;
    ao::launch(kern, dim3((unsigned)(p.N / 8)), dim3(waves * 64), smem, stream, p);
    AO_LAUNCH_CHECK("dec8_kernel launch");
    return AO_OK;
  }
  const void* kern = half ? reinterpret_cast<const void*>(dec8_kernel<INT8, DYN, DEPTH, XFAST, true>)
                          : reinterpret_cast<const void*>(dec8_kernel<INT8, DYN, DEPTH, XFAST, false>);
  if (int rc = ensure_dynamic_lds(kern, smem, "hipFuncSetAttribute(dec8_kernel)")) return rc;
  if (half) ao::launch(dec8_kernel<INT8, DYN, DEPTH, XFAST, true>, dim3((unsigned)(p.N / 16)), dim3(waves * 64), smem, stream, p);
  else ao::launch(dec8_kernel<INT8, DYN, DEPTH, XFAST, false>, dim3((unsigned)(p.N / 16)), dim3(waves * 64), smem, stream, p);
  AO_LAUNCH_CHECK("dec8_kernel launch");
  return AO_OK;
}

template <bool INT8, bool DYN>
int launch_dec8_loop(const Dec8Args& p, int waves, hipStream_t stream) {
  const size_t smem = dec8_lds(p.M, p.K, waves);
  auto kern = dec8_kernel<INT8, DYN, 4, false, false, true>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(dec8_kernel)")) return rc;
  ao::launch(kern, dim3((unsigned)(p.N / 16)), dim3(waves * 64), smem, stream, p);
  AO_LAUNCH_CHECK("dec8_kernel launch");
  return AO_OK;
}

template <bool INT8, bool DYN>
int launch_dec8(const Dec8Args& p, const Dec8Shape& s, hipStream_t stream) {
  if (s.loop) return launch_dec8_loop<INT8, DYN>(p, s.waves, stream);
  const bool xfast = DYN ? (p.M == 1) : (p.M * s.depth * 8 <= kXV * 64);
  const bool half = g_dec8_mode == 290;
#define AO_DEC8_CASE(D)                                                                      \
  case D:                                                                                    \
    return xfast ? launch_dec8_h<INT8, DYN, D, true>(p, s.waves, half, stream) : launch_dec8_h<INT8, DYN, D, false>(p, s.waves, half, stream);
  switch (s.depth) {
    AO_DEC8_CASE(8)
    AO_DEC8_CASE(7)
    AO_DEC8_CASE(4)
    AO_DEC8_CASE(2)
    AO_DEC8_CASE(1)
  }
#undef AO_DEC8_CASE
  set_error("dec8: no instantiation for ring depth %d", s.depth);
  return AO_ERR_INVALID_ARGUMENT;
}

// Whether the straight-line decode kernel takes this problem: M <= 16, the codes of the activation + the slabs within the CU's LDS,
// K = 128 x depth x waves.  A function of the shape only, so that the fused (cast inside) and the two-launch forms of one linear pick
// the same wave split and give the same bits.
bool dec8_takes_shape(int64_t M, int64_t N, int64_t K, Dec8Shape* shape) {
  if (g_dec8_mode == 299) return false;
  if (M < 1 || M > 16 || N % 16 != 0 || K % 128 != 0 || N >= (1ll << 31) || K >= (1ll << 24)) return false;
  Dec8Shape s;
  const int forced = (g_dec8_mode > 200 && g_dec8_mode <= 208) ? g_dec8_mode - 200 : 0;
  if (!dec8_shape(K, forced, &s)) return false;
  // Round 6: 9 .. 16 rows pick the ring by occupancy.  The deepest ring (round 4's rule, measured at M = 1) means the fewest waves per workgroup;
  // with many rows the activation codes cap the workgroups per CU, and on narrow weights there are few workgroups to begin with -- o shard
  // 8192 x 1024 ran ONE wave x 2 workgroups per CU.  Resident waves per CU = waves x min(workgroups LDS admits, workgroups the grid offers);
  // among the rings that do not add a round of the chip: the deepest with >= 8 resident waves (two per SIMD; asking for 12 cost 3 - 7 % on
  // the qkv shard 1280 x 8192 and on 8192 x 3584), else the one with the most.  Measured
  // (profiles/dec8_depth_tuned_r06.jsonl, other_shapes_dec8_r06.jsonl, cold weights, M = 9 / 12 / 16): o shard 8192 x 1024 5.1 / 5.3 / 6.1 ->
  // 4.2 us, o 4096^2 6.3 / 6.6 / 7.2 -> 6.0 / 6.0 / 6.4, 8192 x 2048 6.7 - 7.8 -> 5.5 - 5.8, 3584^2 5.5 - 6.2 -> 5.0 - 5.3, K = 5120 at
  // 14 - 16 rows (5 waves alone on a CU) 24.7 - 25.9 -> 21.9 - 22.3.
  // 5 .. 8 rows: the deepest ring whose activation slice a wave can still hold in registers (XFAST: M x depth <= 32 -- depth 4), where K has one.
  // profiles/dec8_small_m_forms_r06.jsonl (fp8, cold, 20 shapes, M = 6 / 8): o shard 8192 x 1024 4.5 / 4.7 -> 3.9 us, 8192 x 2048 6.1 / 6.3 ->
  // 5.2 / 5.3, qkv shard 1280 x 8192 5.3 / 5.4 -> 4.9 / 5.1, gate_up shard at 8 rows 15.4 -> 14.2 (at 6: 13.4 -> 14.0, the one cell behind),
  // 37888 x 3584 at 8 rows 28.6 -> 25.6; level elsewhere.  Up to 4 rows the deepest ring is ahead on every shape (round 4's rule holds).
  if (forced == 0 && !s.loop && M >= 5 && M <= 8 && M * s.depth > 32 && g_dec8_mode != 293) {
    const int ksteps = (int)(K / 128);
    static const int depths[] = {8, 7, 4, 2, 1};
    for (int d : depths) {
      if (M * d > 32 || ksteps % d != 0) continue;
      const int w = ksteps / d;
      if (w < 1 || w > 16) break;  // shallower rings only add waves
      if (dec8_lds(M, K, w) <= 160 * 1024) s = Dec8Shape{w, d, false};
      break;
    }
  }
  if (forced == 0 && !s.loop && M >= 9 && g_dec8_mode != 293) {
    const int ksteps = (int)(K / 128);
    const int64_t wgs = N / 16, offered = std::max<int64_t>(1, wgs / 256);
    static const int depths[] = {8, 7, 4, 2, 1};
    int best_d = 0, best_res = -1;
    int64_t best_rounds = 0;
    bool settled = false;
    for (int pass = 0; pass < 2 && !settled; ++pass) {  // pass 0: the fewest rounds any ring reaches; pass 1: pick among those
      for (int d : depths) {
        if (ksteps % d != 0) continue;
        const int w = ksteps / d;
        if (w < 1 || w > 16) continue;
        const size_t lds = dec8_lds(M, K, w);
        if (lds > 160 * 1024) continue;
        const int64_t cap = (int64_t)((160 * 1024) / lds), rounds = (wgs + 256 * cap - 1) / (256 * cap);
        const int res = (int)(w * std::min<int64_t>(cap, offered));
        if (pass == 0) {
          if (best_rounds == 0 || rounds < best_rounds) best_rounds = rounds;
          continue;
        }
        if (rounds != best_rounds) continue;
        if (res >= 8) {  // depths run deepest first
          best_d = d;
          settled = true;
          break;
        }
        if (res > best_res) {
          best_res = res;
          best_d = d;
        }
      }
    }
    if (best_d != 0) s = Dec8Shape{ksteps / best_d, best_d, false};
  }
  // Round 6: the activation codes may fill the CU's 160 KiB of LDS (one workgroup per CU from ~80 KiB on).  Rounds 4 - 5 stopped at 64 KiB of
  // codes (the fused-cast kernel's bound; variant 293 keeps it for A/B), which sent M = 8 .. 16 on K = 8192 to the per-tile streaming kernel:
  // qkv shard 1280 x 8192 9.1 - 10.6 us -> 5.6 - 6.4, gate_up shard 19 - 21 -> 15.6 - 17.5, qkv 6144 x 4096 at M = 16 11.8 -> 8.9,
  // down 4096 x 14336 at M = 4 .. 8 16.3 - 17.3 -> 12.8 - 15.7, no cell slower (profiles/dec8_lds_cap_ab_r06.jsonl, cold weights, same bits)
  if (g_dec8_mode == 293) {
    if (dec8_lds(M, K, s.waves) > 64 * 1024 + 24 * 1024 || M * (K + 16) > 64 * 1024) return false;
  } else if (dec8_lds(M, K, s.waves) > 160 * 1024) {
    return false;
  }
  if (shape != nullptr) *shape = s;
  return true;
}

}  // namespace

bool dec8_takes(int64_t M, int64_t N, int64_t K) { return dec8_takes_shape(M, N, K, nullptr); }

// xq . wq^T with the scale epilogue, activation already cast (aten::_int_mm + scales / aten::_scaled_mm rowwise at decode sizes)
int dec8_scaled(bool int8, const void* xq, const float* x_scale, const void* wq, const float* w_scale, const uint16_t* bias, uint16_t* y,
                int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  Dec8Shape s;
  if (!dec8_takes_shape(M, N, K, &s)) {
    set_error("dec8_scaled: shape M=%lld N=%lld K=%lld not covered", (long long)M, (long long)N, (long long)K);
    return AO_ERR_INVALID_ARGUMENT;
  }
  Dec8Args p{xq, x_scale, reinterpret_cast<const uint8_t*>(wq), w_scale, bias, y, (int)M, (int)N, (int)K};
  return int8 ? launch_dec8<true, false>(p, s, stream) : launch_dec8<false, false>(p, s, stream);
}

// the dynamic-activation linear with the cast fused in
int dec8_dynamic(bool int8, const uint16_t* x, const void* wq, const float* w_scale, const uint16_t* bias, uint16_t* y, int64_t M, int64_t N,
                 int64_t K, hipStream_t stream) {
  Dec8Shape s;
  if (!dec8_takes_shape(M, N, K, &s)) {
    set_error("dec8_dynamic: shape M=%lld N=%lld K=%lld not covered", (long long)M, (long long)N, (long long)K);
    return AO_ERR_INVALID_ARGUMENT;
  }
  Dec8Args p{x, nullptr, reinterpret_cast<const uint8_t*>(wq), w_scale, bias, y, (int)M, (int)N, (int)K};
  return int8 ? launch_dec8<true, true>(p, s, stream) : launch_dec8<false, true>(p, s, stream);
}

}  // namespace ao
