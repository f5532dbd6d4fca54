// M == 1 (decode) int4 weight-only GEMV for gfx950, "balanced streaming" form.
//
//     y[1,N] = x[1,K] @ dequant(qdata)^T            (aten::_weight_int4pack_mm, M = 1;
//     call site torchao/quantization/quantize_/workflows/int4/int4_tile_packed_to_4d_tensor.py:287)
//
// Why this shape.  With the oracle's double rounding replayed exactly
// (w = bf16(bf16((q-8)*s) + z), torchao/quantization/quant_primitives.py:999-1007) the kernel is
// instruction-issue bound in steady state (~350 cycles per 1 KiB weight block per SIMD), so wall
// time = blocks on the busiest SIMD x cycles per block + whatever start-up latency is exposed.
//   * grid = one workgroup per CU; workgroup c owns n-tiles perm(c), perm(c)+C, ... (perm keeps 32
//     neighbouring tiles -- they share 128-byte lines of the [K/g][N][2] scale/zero tensor -- on
//     one XCD, i.e. one L2);
//   * the T mod C left-over tiles are cut S = 1, 2 or 4 ways along K over different workgroups
//     ("remainder units", processed FIRST) so that every CU gets T/C tiles' worth of blocks.  Each
//     part publishes 16 fp32 partials as 8-byte {value, tag} granules (one write-through store,
//     no fence, no wait); one wave of the workgroup holding part 0 fetches all parts while its
//     LAST unit is being computed and adds them in part order (deterministic) at the very end;
//   * inside a unit the waves split the unit's k-range; a wave walks the blocks of ALL its units
//     as one stream through a 4-deep register ring of non-temporal 1 KiB wave loads, so the loads
//     of the next tile are in flight while this one is computed;
//   * per-unit split-K sums meet in LDS through an arrival counter; the last arriving wave reduces
//     and stores while the others go on (no workgroup barrier after the x staging).
// The ring is issued from inline asm with hand-counted s_waitcnt vmcnt(N): the compiler's own
// counting answers every wave-uniform branch with vmcnt(0), and its workgroup fences do the same,
// each of which drains the ring (measured with the trace build: +1.4 us per flush).
//
// Weight format: see int4_kernels.hip (bit-exact aten::_convert_weight_to_int4pack layout).
#include "common.h"
#include "lds_dma.h"

#include <mutex>

namespace ao {
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 4;         // weight blocks in flight per wave
constexpr int kRedBufs = 4;      // LDS split-K buffers (units in flight per workgroup), power of two
constexpr int kMaxLogSplit = 2;  // a remainder tile is cut into at most 4 k-ranges (one wave holds 4 x 16 granules)
constexpr int kWsSlots = 32;     // rotating workspace slots (launches in flight per device)
constexpr int kTraceStride = 64; // trace build: u64 stamps per workgroup (6 of wave 0, then 3 per wave)

struct StreamPlan {
  int grid;       // workgroups C
  int full;       // F = tiles / C full tiles per workgroup
  int rem_units;  // R * S remainder units (R = tiles % C left-over tiles, each cut into S k-ranges),
                  // dealt round-robin: workgroup c takes units c, c + C, ...
  int log_split;  // log2 S
  int kblocks;    // K / 128
  int xcd_span;   // C / 8 when the grid is a multiple of 8 (XCD-contiguous tile order), else 0
};

// row `lane & 3` of the 4x4 identity as a 4x4x4 A operand (4 bf16)
__device__ __forceinline__ s16x4 identity_row4(int lane) {
  const int hot = lane & 3;
  s16x4 f;
  f.x = hot == 0 ? (short)0x3F80 : (short)0;
  f.y = hot == 1 ? (short)0x3F80 : (short)0;
  f.z = hot == 2 ? (short)0x3F80 : (short)0;
  f.w = hot == 3 ? (short)0x3F80 : (short)0;
  return f;
}

// D = I * B + C on the matrix pipe: each lane gets its own four bf16 values widened to fp32
// plus c, correctly rounded (one product per output) -- torch's bf16 add before its rounding.
__device__ __forceinline__ f32x4 widen_add4(s16x4 ident, uint32_t lo_pair, uint32_t hi_pair, f32x4 c) {
  const u32x2 bb = {lo_pair, hi_pair};
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ident, __builtin_bit_cast(s16x4, bb), c, 0, 0, 0);
}

// ---- hand-counted memory pipeline ---------------------------------------------------------
// Everything that streams goes global -> LDS by LDS-DMA (global_load_lds_*: the wave writes
// [M0 + lane * size], no VGPR destination) issued from inline asm and retired by explicit
// s_waitcnt vmcnt(N).  Why not plain loads: (a) hipcc's own counting answers every wave-uniform
// branch between issue and use (ring refill at the end of a wave's stream, ring slot chosen at run
// time, a store on one path) with vmcnt(0), which drains the ring; (b) an inline-asm load INTO
// REGISTERS is unsafe -- the register allocator may copy or move the destination before the data
// lands (it did: v_mov of in-flight ring registers at the loop head).  LDS-DMA has no register
// destination, so (b) cannot happen, and the wave's own ds_read after its vmcnt wait sees the data.
// VMEM loads return in order, so "at most N outstanding" == "everything older than the N youngest
// has landed"; an extra store or older load in the pipe only makes such a wait more conservative.
// wait until at most `stages` ring stages (LPS DMAs each) are still in flight
template <int LPS>
__device__ __forceinline__ void wait_ring(int stages) {
  switch (stages) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<LPS * 1>(); break;
    case 2: wait_vmcnt<LPS * 2>(); break;
    case 3: wait_vmcnt<LPS * 3>(); break;
    default: wait_vmcnt<LPS * 4>(); break;
  }
}

// One packed word (8 nibbles) -> four packed-bf16 pairs (k-slots 0-3 = inner tile 2j,
// 4-7 = inner tile 2j+1), the oracle's rounding sequence bit for bit.
//   * a byte holding q (0..15) read as OCP e4m3 is q * 2^-9, so v_cvt_scalef32_pk_f32_fp8
//     with scale 2^9 converts two nibbles per instruction, exactly;
//   * (q-8)*s is exact in fp32; v_cvt_pk_bf16_f32 is rounding #1;
//   * t + z is one IEEE fp32 add (identity MFMA), v_cvt_pk_bf16_f32 is rounding #2.
__device__ __forceinline__ void dequant_word_exact(uint32_t p, float s, float neg8s, f32x4 zz, s16x4 ident,
                                                   uint32_t (&out)[4]) {
  const uint32_t lo = p & 0x0F0F0F0Fu;         // bytes: v0, v4, v1, v5
  const uint32_t hi = (p >> 4) & 0x0F0F0F0Fu;  // bytes: v2, v6, v3, v7
  const f32x2 r0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, false);  // q0, q4
  const f32x2 r1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, 512.0f, true);   // q1, q5
  const f32x2 r2 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, false);  // q2, q6
  const f32x2 r3 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hi, 512.0f, true);   // q3, q7
  const f32x2 t0 = r0 * s + neg8s, t1 = r1 * s + neg8s, t2 = r2 * s + neg8s, t3 = r3 * s + neg8s;
  const uint32_t tp0 = pack_bf16x2(t0.x, t1.x), tp1 = pack_bf16x2(t2.x, t3.x);
  const uint32_t tp2 = pack_bf16x2(t0.y, t1.y), tp3 = pack_bf16x2(t2.y, t3.y);
  const f32x4 w0 = widen_add4(ident, tp0, tp1, zz);
  const f32x4 w1 = widen_add4(ident, tp2, tp3, zz);
  out[0] = pack_bf16x2(w0.x, w0.y); out[1] = pack_bf16x2(w0.z, w0.w);
  out[2] = pack_bf16x2(w1.x, w1.y); out[3] = pack_bf16x2(w1.z, w1.w);
}

// MODE (profiling builds only): 0 = product, 1 = loads but no dequant/MFMA, 3 = product + six
// s_memrealtime stamps (100 MHz) of wave 0 per workgroup: entry, x landed, staging barrier passed,
// first block consumed, last block consumed, exit
template <int G, int LOGW, int MODE = 0>
__global__ __launch_bounds__(64 << LOGW) void int4_gemv_stream_kernel(
    const uint16_t* __restrict__ x, const u32x4* __restrict__ qdata, const uint32_t* __restrict__ sz,
    uint16_t* __restrict__ y, int N, int K, StreamPlan plan, u32x4* __restrict__ ws,
    unsigned long long* __restrict__ trace) {
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
  if (MODE == 3) ts[0] = __builtin_amdgcn_s_memrealtime();
  constexpr int NG = (G >= 128) ? 1 : (128 / G);  // scale/zero words per 128-k block
  constexpr int W = 1 << LOGW;                    // waves per workgroup
  constexpr int LPS = 1 + NG;                     // DMAs per ring stage
  constexpr int STAGE = 1024 + NG * 256;          // bytes: one 1 KiB weight block + NG x 64 scale/zero words
  static_assert(kRing == 4 && kRedBufs == 4, "wait_ring and the buffer index are written for 4");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [W][ring | x slices | zero row], [kRedBufs][W][16] f32 partials, arrival counters, generations,
  //      [64] granule landing pad

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x;
  const int kq = lane >> 4;
  const int nl = lane & 15;
  const int S1 = (1 << plan.log_split) - 1;

  // per wave: [kRing stages][x slices of this wave's k-ranges, A-fragment order][64 B of zeros]
  const int xw_blocks = 2 * ((plan.kblocks + W - 1) / W) + 2;  // full-tile range + remainder range, upper bound (host uses the same)
  const int wave_bytes = kRing * STAGE + xw_blocks * 256 + 64;
  char* ring = smem + wave * wave_bytes;
  char* xw = ring + kRing * STAGE;
  char* zero_row = xw + xw_blocks * 256;
  float* red = reinterpret_cast<float*>(smem + W * wave_bytes);
  uint32_t* arrive = reinterpret_cast<uint32_t*>(red + kRedBufs * W * 16);
  uint32_t* gen = arrive + kRedBufs;
  u32x4* pad = reinterpret_cast<u32x4*>(gen + kRedBufs);  // [64] granules {partial bits, tag, 0, 0} of the owned K-split tile (16-byte aligned: all sizes before it are)
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(lds_offset(ring));

  // ---- units of this workgroup: remainder units c, c + C, ... first, then F full tiles -------
  int nrem = 0;
#pragma nounroll
  for (int u = c; u < plan.rem_units; u += plan.grid) ++nrem;
  const int nunits = plan.full + nrem;
  const int my_tile0 = plan.xcd_span ? ((c & 7) * plan.xcd_span + (c >> 3)) : c;
  const int rem_tile0 = plan.full * plan.grid;
  // unit -> tile and this wave's block range [b0, b1)
  auto unit_range = [&](int ui, int& tile, int& b0, int& b1) {
    int u0 = 0, u1 = plan.kblocks;
    if (ui < nrem) {
      const int u = c + ui * plan.grid;
      const int part = u & S1;
      tile = rem_tile0 + (u >> plan.log_split);
      u0 = (plan.kblocks * part) >> plan.log_split;
      u1 = (plan.kblocks * (part + 1)) >> plan.log_split;
    } else {
      tile = (ui - nrem) * plan.grid + my_tile0;
    }
    const int len = u1 - u0;
    b0 = u0 + ((len * wave) >> LOGW);
    b1 = u0 + ((len * (wave + 1)) >> LOGW);
  };

  // ---- 1a. counters and zero row; the only workgroup barrier of the kernel (nothing is in flight yet)
  if (tid < 2 * kRedBufs) arrive[tid] = 0u;  // arrive[] and gen[] are contiguous
  if (lane < 16) reinterpret_cast<uint32_t*>(zero_row)[lane] = 0u;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- 1b. x staging, wave-private: each wave DMAs exactly the x slices of ITS k-ranges (one for
  // the full tiles, one for its remainder unit) -- 256 B per 128-k block, one 4-byte-per-lane DMA
  // each -- AHEAD of its own weight ring.  Loads return in order, so when the first weight stage
  // has landed x has too: no staging barrier, no wave waits for another wave's x, and the waves'
  // first blocks start staggered instead of all at once.  (A workgroup-wide staging needed two
  // barriers: one so that no wave's 4 KiB ring prologue queued ahead of another wave's x request
  // in the CU's in-order memory pipe, one for the hand-over; trace: 0.7 us from x landed to barrier.)
  // LDS order per block [kq][j][8]: k = 32 j + 16 h + 4 kq + i sits at slot 4 h + i, so the A operand
  // of MFMA j is one ds_read_b128 at block + kq*64 + j*16.
  int fb0, fb1, rb0 = 0, rb1 = 0;  // this wave's block ranges: full tiles, remainder unit
  {
    fb0 = (plan.kblocks * wave) >> LOGW;
    fb1 = (plan.kblocks * (wave + 1)) >> LOGW;
    if (plan.full == 0) fb1 = fb0;
    if (nrem > 0) {
      int t;
      unit_range(0, t, rb0, rb1);
    }
    const uint32_t xw_lds = __builtin_amdgcn_readfirstlane(lds_offset(xw));
    const int d = lane >> 1;  // 8-byte dest unit within the block: kq*8 + j*2 + h
    const int src4 = 2 * (8 * ((d >> 1) & 3) + 4 * (d & 1) + (d >> 3)) + (lane & 1);  // source 4-byte unit within the block
    const uint16_t* xl = x + src4 * 2;
#pragma nounroll
    for (int kb = fb0; kb < fb1; ++kb) dma_b32(xl + (size_t)kb * 128, xw_lds + (kb - fb0) * 256);
#pragma nounroll
    for (int kb = rb0; kb < rb1; ++kb) dma_b32(xl + (size_t)kb * 128, xw_lds + (fb1 - fb0 + kb - rb0) * 256);
  }

  // ---- 1b. weight ring prologue: prefetch cursor over this wave's block stream
  int pf_ui = -1, pf_kb = 0, pf_b1 = 0, pf_tile = 0;
  bool pf_valid = true;
  auto pf_advance = [&]() {
    ++pf_kb;
#pragma nounroll
    while (pf_kb >= pf_b1) {
      if (++pf_ui >= nunits) {
        pf_valid = false;
        return;
      }
      unit_range(pf_ui, pf_tile, pf_kb, pf_b1);
    }
  };
  pf_advance();
  auto issue = [&](int slot) {
    const uint32_t dst = ring_lds + slot * STAGE;
    dma_b128_nt(qdata + ((size_t)pf_tile * plan.kblocks + pf_kb) * 64 + lane, dst);
    const int kg0 = (G >= 128) ? ((pf_kb * 128) / G) : (pf_kb * NG);
#pragma unroll
    for (int i = 0; i < NG; ++i) dma_b32(sz + (size_t)(kg0 + i) * N + pf_tile * 16 + nl, dst + 1024 + i * 256);
    pf_advance();
  };
  int inflight = 0;  // ring stages issued and not yet consumed
#pragma nounroll
  for (int d = 0; d < kRing; ++d) {
    if (pf_valid) {
      issue(d);
      ++inflight;
    }
  }

  // ---- 3. main loop
  const bool row0 = nl == 0;  // lanes holding row 0 (= x) of the 16x16 MFMA tile; the rest read zeros
  const char* a_lane = row0 ? xw + kq * 64 : zero_row;  // + block slot * a_stride
  const int a_stride = row0 ? 256 : 0;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const s16x4 ident = identity_row4(lane);

  // granules of the K-split tile this workgroup owns (part 0 of its remainder unit; with S <= 4 one
  // wave holds all S x 16 of them): fetched by wave W-1 while its last unit is computed
  const bool owner = nrem > 0 && plan.log_split > 0 && (c & S1) == 0;
  u32x4* gran_ptr = ws + ((size_t)(c >> plan.log_split) << plan.log_split) * 16 + lane;
  const uint32_t pad_lds = __builtin_amdgcn_readfirstlane(lds_offset(pad));
  auto fetch_granules = [&]() { dma_b128_sc1(gran_ptr, pad_lds); };

  int slot = 0;
  bool first = true;
#pragma nounroll
  for (int ui = 0; ui < nunits; ++ui) {
    int tile, b0, b1;
    unit_range(ui, tile, b0, b1);
    const int xbase = (ui < nrem) ? rb0 - (fb1 - fb0) : fb0;  // block kb's x sits at slot kb - xbase
    if (owner && wave == W - 1 && ui == nunits - 1) fetch_granules();
#pragma nounroll
    for (int kb = b0; kb < b1; ++kb) {
      wait_ring<LPS>(inflight - 1);  // the oldest stage has landed (and with it this wave's x); the younger ones may be in flight
      if (MODE == 3 && first) ts[1] = ts[2] = __builtin_amdgcn_s_memrealtime();
      const char* sb = ring + slot * STAGE;
      const u32x4 wv = *reinterpret_cast<const u32x4*>(sb + lane * 16);
      uint32_t szv[NG];
#pragma unroll
      for (int i = 0; i < NG; ++i) szv[i] = *reinterpret_cast<const uint32_t*>(sb + 1024 + i * 256 + lane * 4);
      if (MODE == 1) {
        acc.x += bits_to_f32((wv.x ^ wv.y ^ wv.z ^ wv.w ^ szv[0]) & 0x3f800000u);
      } else {
        const uint32_t wds[4] = {wv.x, wv.y, wv.z, wv.w};
        u32x4 bw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gi = (G >= 128) ? 0 : ((j * 32) / G);
          const float sc = bf16_lo_to_f32(szv[gi]);
          const float zp = bf16_hi_to_f32(szv[gi]);
          uint32_t b[4];
          dequant_word_exact(wds[j], sc, -8.0f * sc, f32x4{zp, zp, zp, zp}, ident, b);
          bw[j] = u32x4{b[0], b[1], b[2], b[3]};
        }
        const char* a_ptr = a_lane + (kb - xbase) * a_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(a_ptr + j * 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                        __builtin_bit_cast(bf16x8, bw[j]), acc, 0, 0, 0);
        }
      }
      // The slot's ds_reads have returned (their data fed the VALU above; the compiler's lgkmcnt
      // wait precedes those uses), so the slot may be refilled; or the ring runs dry at the end.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (pf_valid) issue(slot);
      else --inflight;
      slot = (slot + 1) & (kRing - 1);
      if (MODE == 3 && first) { ts[3] = __builtin_amdgcn_s_memrealtime(); first = false; }
    }
    if (MODE == 3 && ui == nunits - 1) ts[4] = __builtin_amdgcn_s_memrealtime();

    // ---- split-K meeting point of unit ui: every wave of the workgroup arrives once.
    // LDS operations of one wave execute in order and the compiler waits (lgkmcnt) for returned
    // data, so the only thing to prevent is compile-time reordering -- no fences (they would add
    // vmcnt(0) and drain the ring).
    const int b = ui & (kRedBufs - 1);
    const uint32_t want = (uint32_t)(ui >> 2);
    while (__hip_atomic_load(&gen[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != want) __builtin_amdgcn_s_sleep(1);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (lane < 16) red[(b * W + wave) * 16 + lane] = acc.x;
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    uint32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&arrive[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (old == (uint32_t)(W - 1)) {
      // last arriver: reduce in wave order, free the buffer, store
      float sum = 0.f;
      if (lane < 16) {
#pragma unroll
        for (int w = 0; w < W; ++w) sum += red[(b * W + w) * 16 + lane];
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      if (lane == 0) {
        __hip_atomic_store(&arrive[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&gen[b], want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (ui >= nrem || plan.log_split == 0) {
        if (lane < 16) y[tile * 16 + lane] = f32_to_bf16_bits(sum);
      } else if (lane < 16) {
        // K-split tile: publish {partial, tag} -- one 16-byte write-through store per lane, no wait
        const int u = c + ui * plan.grid;
        store_b128_sc1(ws + (size_t)u * 16 + lane, u32x4{f32_to_bits(sum), 1u, 0u, 0u});
      }
    }
  }

  // ---- 4. the owner adds the parts of its K-split tile (fixed order) and writes y
  if (owner && wave == W - 1) {
    const int nl_gran = 16 << plan.log_split;
    // by now the partners' granules are microseconds old; poll only if a tag is missing
    uint32_t gval = 0;
    for (;;) {
      wait_vmcnt<0>();  // the granule fetch (and everything else this wave issued) has landed
      const uint32_t* pw = reinterpret_cast<const uint32_t*>(pad + lane);
      gval = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t tag = __hip_atomic_load(pw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (__ballot(lane < nl_gran && tag != 1u) == 0ull) break;
      __builtin_amdgcn_s_sleep(8);
      fetch_granules();
    }
    const float v = bits_to_f32(gval);
    float t = v;  // part 0, then 1, 2, 3
    for (int p = 1; p <= S1; ++p) t += __shfl(v, nl + 16 * p);
    const int tile = rem_tile0 + (c >> plan.log_split);
    if (lane < 16) y[tile * 16 + lane] = f32_to_bf16_bits(t);
    if (lane < nl_gran) store_b128_sc1(gran_ptr, u32x4{0u, 0u, 0u, 0u});  // tags back to "empty" for the next launch
  }
  if (MODE == 3 && trace != nullptr && lane == 0) {
    ts[5] = __builtin_amdgcn_s_memrealtime();
    unsigned long long* t = trace + (size_t)c * kTraceStride;
    if (wave == 0) for (int i = 0; i < 6; ++i) t[i] = ts[i];
    t[8 + 3 * wave + 0] = ts[3];
    t[8 + 3 * wave + 1] = ts[4];
    t[8 + 3 * wave + 2] = ts[5];
  }
}

// ---- host side ------------------------------------------------------------------------

struct DeviceState {
  int cus = 0;
  u32x4* ws = nullptr;  // [kWsSlots][(cus + 4) * 16] granules {fp32 partial, tag, 0, 0}; tags start and end at 0
  unsigned next_slot = 0;
};
std::mutex g_mu;
DeviceState g_dev[64];

int device_state(DeviceState** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_failed(e, "hipGetDevice");
  AO_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  DeviceState& d = g_dev[dev];
  if (d.ws == nullptr) {
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return hip_failed(e, "hipDeviceGetAttribute(MultiprocessorCount)");
    AO_REQUIRE(cus > 0, "device reports %d compute units", cus);
    u32x4* ws = nullptr;
    const size_t bytes = (size_t)kWsSlots * (cus + 4) * 16 * sizeof(u32x4);  // +4: the owner's 64-lane fetch of the last tile
    e = hipMalloc(&ws, bytes);
    if (e != hipSuccess)
      return hip_failed(e, "hipMalloc(int4 split-K workspace); call ao_int4_weight_int4pack_mm once outside "
                           "stream capture before capturing it into a graph");
    e = hipMemset(ws, 0, bytes);
    if (e != hipSuccess) { (void)hipFree(ws); return hip_failed(e, "hipMemset(int4 split-K workspace)"); }
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(ws); return hip_failed(e, "hipDeviceSynchronize(int4 split-K workspace)"); }
    d.cus = cus;
    d.ws = ws;
  }
  *out = &d;
  return AO_OK;
}

int g_stream_wpb = 16;
unsigned long long* g_trace = nullptr;  // profiling only (ao_int4_set_trace)

// Unit plan for T tiles on C compute units (see the header comment).
StreamPlan make_plan(int tiles, int kblocks, int cus) {
  StreamPlan p;
  p.kblocks = kblocks;
  p.log_split = 0;
  int rem;
  if (tiles >= cus) {
    p.grid = cus;
    p.full = tiles / cus;
    rem = tiles % cus;
  } else {
    // fewer tiles than CUs: every tile is a "left-over" tile, cut along K to fill the chip
    p.grid = cus;
    p.full = 0;
    rem = tiles;
  }
  if (rem > 0) {
    // S in {1, 2, 4} with R*S <= C (each workgroup at most one remainder unit, so the parts of a tile
    // sit in S distinct workgroups), minimising the busiest workgroup's share: 1/S
    for (int ls = 1; ls <= kMaxLogSplit && (1 << ls) <= kblocks && (rem << ls) <= p.grid; ++ls) p.log_split = ls;
  }
  p.rem_units = rem << p.log_split;
  if (p.full == 0 && p.rem_units < p.grid) p.grid = p.rem_units;  // no idle workgroups
  p.xcd_span = (p.grid % 8 == 0) ? p.grid / 8 : 0;
  return p;
}

template <int G, int LOGW, int MODE>
int launch_stream_impl(const uint16_t* x, const int32_t* qdata, const uint16_t* sz, uint16_t* y, int64_t N,
                       int64_t K, hipStream_t stream) {
  DeviceState* d = nullptr;
  unsigned slot = 0;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (int rc = device_state(&d)) return rc;
    slot = d->next_slot++ % kWsSlots;
  }
  const StreamPlan p = make_plan((int)(N >> 4), (int)(K >> 7), d->cus);
  constexpr int wpb = 1 << LOGW;
  constexpr int stage = 1024 + ((G >= 128) ? 1 : (128 / G)) * 256;
  const int xw_blocks = 2 * ((p.kblocks + wpb - 1) / wpb) + 2;
  const size_t smem = (size_t)wpb * (kRing * stage + xw_blocks * 256 + 64) + (size_t)kRedBufs * wpb * 16 * sizeof(float) +
                      2 * kRedBufs * sizeof(uint32_t) + 64 * sizeof(u32x4);
  auto kern = int4_gemv_stream_kernel<G, LOGW, MODE>;
  if (smem > 48 * 1024) {
    static size_t granted = 0;  // monotonic; a racing duplicate call is harmless
    if (smem > granted) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return hip_failed(e, "hipFuncSetAttribute(int4_gemv_stream_kernel)");
      granted = smem;
    }
  }
  ao::launch(kern, dim3((unsigned)p.grid), dim3(wpb * 64), smem, stream, x, reinterpret_cast<const u32x4*>(qdata),
             reinterpret_cast<const uint32_t*>(sz), y, (int)N, (int)K, p, d->ws + (size_t)slot * (d->cus + 4) * 16, g_trace);
  AO_LAUNCH_CHECK("int4_gemv_stream_kernel launch");
  return AO_OK;
}

}  // namespace

// per wave: the ring (5 KiB at g >= 128, 8 KiB at g = 32) + its x slices (2 * ceil(K/128/W) + 2 blocks of
// 256 B); 16 (8 at g < 128) waves + 5 KiB of scratch must fit the 160 KiB LDS of a CU
bool int4_gemv_stream_supported(int64_t K) { return K <= 16384; }

void int4_gemv_stream_set_trace(unsigned long long* p) { g_trace = p; }

void int4_gemv_stream_set_waves(int waves) { g_stream_wpb = (waves == 8) ? 8 : 16; }

int launch_int4_gemv_stream(const uint16_t* x, const int32_t* qdata, const uint16_t* sz, uint16_t* y, int64_t N,
                            int64_t K, int group_size, int mode, hipStream_t stream) {
#define AO_STREAM_CASE(G)                                                                         \
  case G:                                                                                         \
    if (mode == 1) return launch_stream_impl<G, 4, 1>(x, qdata, sz, y, N, K, stream);             \
    if (mode == 3) return launch_stream_impl<G, 4, 3>(x, qdata, sz, y, N, K, stream);             \
    if (g_stream_wpb == 8 || G < 128) return launch_stream_impl<G, 3, 0>(x, qdata, sz, y, N, K, stream); \
    return launch_stream_impl<G, 4, 0>(x, qdata, sz, y, N, K, stream);
  switch (group_size) {
    AO_STREAM_CASE(32)
    AO_STREAM_CASE(64)
    AO_STREAM_CASE(128)
    AO_STREAM_CASE(256)
    default:
      set_error("int4 gemv: unsupported group size %d", group_size);
      return AO_ERR_INVALID_ARGUMENT;
  }
#undef AO_STREAM_CASE
}

}  // namespace ao
