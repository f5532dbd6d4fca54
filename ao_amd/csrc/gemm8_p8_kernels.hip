// 1-byte-operand GEMM for large M on gfx950, phase-interleaved 256 x 256 tile: int8 x int8 -> int32 (aten::_int_mm + the
// Int8Tensor scale epilogue, int8/kernels.py:114-144, int8_tensor.py:305-359) and e4m3 x e4m3 -> fp32 (aten::_scaled_mm rowwise,
// float8/inference.py:104-123).  "NT": a [M][K], b [N][K], both K-contiguous.
//
// Structure (the CDNA4 guide's 8-phase idea re-derived for byte operands; one workgroup per CU):
//   * 8 waves = 2 (m) x 4 (n); a wave owns 128 x 64 outputs = 8 x 4 MFMA tiles of 16 x 16 (128 accumulator VGPRs), visited as
//     four QUADRANTS of 64 x 32 per 128-byte K tile: q0 (m-lo, n-lo), q1 (m-lo, n-hi), q2 (m-hi, n-hi), q3 (m-hi, n-lo).  Its B
//     fragments (all 64 columns) and the m-lo A fragments are read in q0, the m-hi A fragments in q2 (16, 0, 8, 0 ds_read_b128).
//   * a PHASE = [LDS fragment reads + 2 LDS-DMA issues + counted vmcnt | barrier | 16 int8 (8 fp8) MFMAs = 256 matrix cycles |
//     barrier].  The two wave rows run one barrier apart, so on every SIMD one wave multiplies while the other reads / issues:
//     the matrix pipe never waits for LDS, and no wave drains its DMA queue in the steady state (vmcnt(4) once per K tile).
//   * operands are staged by LDS-DMA in HALF tiles (128 rows x 128 B = 16 KiB, two 1 KiB DMA instructions per wave), two K
//     tiles of four half tiles resident (128 KiB).  A wave row reads its A half tile in q0 and q2, a wave column pair its B
//     half tile in q0 only, so per K tile t the phases issue: q0 A-lo(t+1), q1 A-hi(t+1), q2 B-lo(t+2), q3 B-hi(t+2) -- every slot
//     has then been unread for two phases when it is refilled (with the one-barrier stagger a slot's last reader can be a phase
//     behind its writer), and the wait for tile t+1 (vmcnt(4) in q3, one phase before its first read: the other wave row's
//     wait is a barrier later) finds its youngest half tile two phases old.
//   * bank conflicts: the DMA writes lane-linear, so the SOURCE is swizzled (LDS chunk position c of row r holds global chunk
//     c ^ ((r >> 1) & 7)) and fragment reads apply the same involution.
//   * epilogue: scales applied in registers with the reference's rounding sequence, tile transposed through the (now free)
//     LDS so that every lane stores 16 contiguous bytes (128-byte rows per 8 lanes) instead of 2.
#include "common.h"
#include "lds_dma.h"
#include "splitk.h"

namespace ao {
namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

enum P8Epi { P8_INT8_SCALED = 0, P8_INT32 = 1, P8_FP8_ROWWISE = 2, P8_FP8_RAW = 3 };

struct P8Args {
  const uint8_t* a;        // [M][K]
  const uint8_t* b;        // [N][K]
  const float* row_scale;  // [M]
  const float* col_scale;  // [N]
  const uint16_t* bias;    // [N] bf16 or null
  void* out;               // bf16 / int32 / fp32 [M][N]
  int M, N, K;
  int tiles_m, tiles_n;
  int group_rows;  // tile rows an XCD's consecutive workgroups walk together (4; ao_gemm8_set_tuning(4, v) for A/B)
  int split;       // K parts per output tile (round 5): the parts meet through split_k_meet2, the last arriver runs the epilogue
  float* ws;
  unsigned* tickets;
};

thread_local int g_p8_group_rows = 0;  // 0 = 4 (product)
thread_local int g_p8h_form = 0;      // lab: loop forms of gemm8_p8h_kernel (fp8 rowwise only)
thread_local int g_p8_split = 0;       // 0 = by shape (p8_split below), n = n K parts wherever they fit
constexpr int kHalf = 16384;          // one half tile: 128 rows x 128 B
constexpr int kBuf = 4 * kHalf;       // A-lo, A-hi, B-lo, B-hi of one K tile
constexpr int kEpiStride = 144;       // bytes per row of a wave's 128 x 64 bf16 staging region (128 + 16: conflict-free b16 writes)
constexpr int kSmem = 8 * 128 * kEpiStride;  // 147456 B >= 2 * kBuf

template <int EPI>
__global__ __launch_bounds__(512) void gemm8_p8_kernel(P8Args p) {
  constexpr bool IS_INT = (EPI == P8_INT8_SCALED || EPI == P8_INT32);
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nl = lane & 15, kq = lane >> 4;

  // workgroup -> tile: blocks of one XCD (id % 8) take consecutive tiles, tiles ordered in groups of GR (= 4: round-5 sweep) tile rows that walk
  // the N direction together (a group shares its A panels in the XCD's L2 and streams B once)
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  // (K parts -- built in round 5 for this tile shape too, never dispatched: a 256 x 256 fp32 partial is 256 KiB, the meeting cost what the
  // shorter loop saved, profiles/p8_split_sweep_r05.jsonl -- live on in the 256 x 128 form below only)
  const int GR = p.group_rows;
  const int group = GR * p.tiles_n;
  const int g0 = (wg / group) * GR;
  const int gsz = min(GR, p.tiles_m - g0);
  const int tm = g0 + (wg % group) % gsz, tn = (wg % group) / gsz;
  const int m0 = tm * 256, n0 = tn * 256;
  const int kb = 0, ktiles = p.K >> 7;

  // DMA sources.  Half tile rows 16 w + 8 i + (lane >> 3), i = 0, 1; lane lands at chunk position lane & 7.
  uint32_t aoff[2][2], boff[2][2];  // [half][i]: byte offset from the tile's first row, k = 0
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 16 * wave + 8 * i + (lane >> 3);
      const uint32_t chunk = (uint32_t)(((lane & 7) ^ (row >> 1)) & 7) << 4;
      aoff[h][i] = (uint32_t)(min(m0 + h * 128 + row, p.M - 1) - m0) * (uint32_t)p.K + chunk;
      boff[h][i] = (uint32_t)(min(n0 + h * 128 + row, p.N - 1) - n0) * (uint32_t)p.K + chunk;
    }
  const uint8_t* abase = p.a + (size_t)m0 * p.K + (size_t)kb * 128;
  const uint8_t* bbase = p.b + (size_t)n0 * p.K + (size_t)kb * 128;
  const uint32_t lds0 = lds_offset(smem);
  // half tile j of the stream: tile j / 4; order A-lo, B-hi, B-lo, A-hi; slots [A-lo, A-hi, B-lo, B-hi]
  auto issue = [&](int tile, int which) {  // which: 0 A-lo, 1 B-hi, 2 B-lo, 3 A-hi
    if (tile >= ktiles) return;
    const bool is_a = (which == 0 || which == 3);
    const int half = (which == 1 || which == 3) ? 1 : 0;
    const uint32_t dst = lds0 + (tile & 1) * kBuf + ((is_a ? 0 : 2) + half) * kHalf + wave * 2048;
    const uint8_t* src = (is_a ? abase : bbase) + (size_t)tile * 128;
    if (is_a) {
      dma_b128_s(src, aoff[half][0], dst);
      dma_b128_s(src, aoff[half][1], dst + 1024);
    } else {
      dma_b128_s(src, boff[half][0], dst);
      dma_b128_s(src, boff[half][1], dst + 1024);
    }
  };

  // fragment addresses inside a buffer: A rows of wave row wr (half tile wr), B rows of wave column wc (half tile wc >> 1)
  const int pos_lo = ((kq ^ (nl >> 1)) & 7) << 4;  // chunk kq of row nl (+ 16 row steps keep (row >> 1) & 7); chunk 4 + kq: ^ 64
  const int a_frag = wr * kHalf + nl * 128 + pos_lo;
  const int b_frag = (2 + (wc >> 1)) * kHalf + ((wc & 1) * 64 + nl) * 128 + pos_lo;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 af[4][2], bf[4][2];  // A: the current 64-row half of the wave's rows (4 m-tiles x {chunk kq, chunk 4 + kq}); B: all 4 n-tiles

  auto load_a = [&](const char* buf, int mi) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int off = a_frag + (mi * 64 + mt * 16) * 128;
      af[mt][0] = *reinterpret_cast<const u32x4*>(buf + off);
      af[mt][1] = *reinterpret_cast<const u32x4*>(buf + (off ^ 64));
    }
  };
  auto load_b = [&](const char* buf) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int off = b_frag + (nt * 16) * 128;
      bf[nt][0] = *reinterpret_cast<const u32x4*>(buf + off);
      bf[nt][1] = *reinterpret_cast<const u32x4*>(buf + (off ^ 64));
    }
  };
  const int unit_scale = 127;  // E8M0 1.0
  auto multiply = [&](int mi, int nj) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int nt = nj * 2 + n2;
        f32x4& c = acc[mi * 4 + mt][nt];
        if constexpr (IS_INT) {
          // (the second k half of every tile follows in a second sweep below: back-to-back MFMAs on one accumulator would wait for each other)
          c = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af[mt][0]), __builtin_bit_cast(i32x4, bf[nt][0]),
                                                                             __builtin_bit_cast(i32x4, c), 0, 0, 0));
        } else {
          const i32x8 av = {(int)af[mt][0].x, (int)af[mt][0].y, (int)af[mt][0].z, (int)af[mt][0].w,
                            (int)af[mt][1].x, (int)af[mt][1].y, (int)af[mt][1].z, (int)af[mt][1].w};
          const i32x8 bv = {(int)bf[nt][0].x, (int)bf[nt][0].y, (int)bf[nt][0].z, (int)bf[nt][0].w,
                            (int)bf[nt][1].x, (int)bf[nt][1].y, (int)bf[nt][1].z, (int)bf[nt][1].w};
          // As inline asm, not the builtin: a builtin MFMA is a pure value computation and the instruction selector is free to place
          // it anywhere in the loop body -- it put all 32 at the END of the K tile (round 3 disassembly: "rrrr DD | | DD | | rrrr DD |
          // | DD MMMM...M |"), so both wave rows multiplied at the same time and loaded at the same time, and the fp8 GEMM ran 21 %
          // behind its int8 twin.  Volatile asm keeps its place between the seams.  (Unit E8M0 scales in a VGPR; cbsz = blgp = 0: e4m3.
          // The accumulator's next reader is the asm MFMA one K tile later or the epilogue: no hazard the compiler has to see.)
          asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(av), "v"(bv), "v"(unit_scale));
        }
      }
    if constexpr (IS_INT) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int nt = nj * 2 + n2;
          f32x4& c = acc[mi * 4 + mt][nt];
          c = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af[mt][1]), __builtin_bit_cast(i32x4, bf[nt][1]),
                                                                             __builtin_bit_cast(i32x4, c), 0, 0, 0));
        }
    }
  };
  // the barrier between a load segment and a multiply segment (and back): LDS reads retired, nothing moves across
  auto seam = [&] {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: K tile 0 entirely and the B halves of tile 1 (as if issued in q2 / q3 of a tile -1); wait for tile 0
  issue(0, 0); issue(0, 3); issue(0, 2); issue(0, 1); issue(1, 2); issue(1, 1);
  if (ktiles > 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
  asm volatile("s_barrier" ::: "memory");
  if (wr == 1) asm volatile("s_barrier" ::: "memory");  // the stagger: wave row 1 runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0; t < ktiles; ++t) {
    const char* buf = smem + (t & 1) * kBuf;
    // ---- q0: m-lo x n-lo; reads the wave's B fragments and its m-lo A fragments; refills A-lo of the other buffer ----------
    load_b(buf); __builtin_amdgcn_sched_barrier(0); load_a(buf, 0);
    issue(t + 1, 0);
    seam();
    __builtin_amdgcn_s_setprio(1); multiply(0, 0); __builtin_amdgcn_s_setprio(0);
    seam();
    // ---- q1: m-lo x n-hi ------------------------------------------------------------------------------------------------
    issue(t + 1, 3);
    seam();
    __builtin_amdgcn_s_setprio(1); multiply(0, 1); __builtin_amdgcn_s_setprio(0);
    seam();
    // ---- q2: m-hi x n-hi; B of this buffer was last read two phases ago: refill B-lo for tile t + 2 ---------------------------
    load_a(buf, 1);
    issue(t + 2, 2);
    seam();
    __builtin_amdgcn_s_setprio(1); multiply(1, 1); __builtin_amdgcn_s_setprio(0);
    seam();
    // ---- q3: m-hi x n-lo; tile t + 1 must have landed before the next phase reads it (its B halves are older than its A halves)
    issue(t + 2, 1);
    if (t + 2 < ktiles) wait_vmcnt<4>(); else wait_vmcnt<0>();
    seam();
    __builtin_amdgcn_s_setprio(1); multiply(1, 0); __builtin_amdgcn_s_setprio(0);
    seam();
  }
  if (wr == 0) asm volatile("s_barrier" ::: "memory");  // wave row 0 catches up: from here on the LDS is free for every wave
  if constexpr (!IS_INT) {
    // the fp8 MFMAs are opaque volatile asm, so the compiler's hazard recognizer does not know that the accumulators are
    // matrix-pipe results still in flight: gfx9 wants up to 19 wait states (16-pass XDL op) between an MFMA's issue and a VALU / VMEM
    // read of its destination, and nothing but luck (the barrier above on one wave row, the epilogue's scale loads) stood between the
    // last multiply() and the epilogue.  20 wait states, explicitly; tests/test_isa_structure.py checks they survive.
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue --------------------------------------------------------------------------------------------------------
  // D layout of the 16 x 16 MFMA: lane (col = nl, kq) holds rows 4 kq + {0..3}
  const int rbase = m0 + wr * 128, cbase = n0 + wc * 64;
  if constexpr (EPI == P8_INT32 || EPI == P8_FP8_RAW) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = rbase + mt * 16 + kq * 4 + r, n = cbase + nt * 16 + nl;
          if (m < p.M && n < p.N) reinterpret_cast<uint32_t*>(p.out)[(size_t)m * p.N + n] = __builtin_bit_cast(u32x4, acc[mt][nt])[r];
        }
  } else {
    float cs[4], bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = min(cbase + nt * 16 + nl, p.N - 1);
      cs[nt] = p.col_scale[n];
      bias[nt] = p.bias != nullptr ? bf16_lo_to_f32(p.bias[n]) : 0.f;
    }
    char* region = smem + wave * (128 * kEpiStride);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      float rs[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[r] = p.row_scale[min(rbase + mt * 16 + kq * 4 + r, p.M - 1)];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if constexpr (EPI == P8_INT8_SCALED) {
            // t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))   (int8_tensor.py:315-359)
            v = round_bf16((float)__builtin_bit_cast(i32x4, acc[mt][nt])[r] * rs[r]) * cs[nt];
          } else {
            v = acc[mt][nt][r] * rs[r] * cs[nt];
          }
          if (p.bias != nullptr) v += bias[nt];
          *reinterpret_cast<uint16_t*>(region + (mt * 16 + kq * 4 + r) * kEpiStride + (nt * 16 + nl) * 2) = f32_to_bf16_bits(v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own writes (DS ops of a wave complete in order)
    uint16_t* out = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = i * 8 + (lane >> 3), piece = lane & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * kEpiStride + piece * 16);
      const int m = rbase + row, n = cbase + piece * 8;
      if (m < p.M && n + 8 <= p.N) *reinterpret_cast<u32x4*>(out + (size_t)m * p.N + n) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// gemm8_p8p_kernel (round 6): the PERSISTENT form of gemm8_p8_kernel for problems with more 256 x 256 tiles than CUs (BASELINE config 3:
// 1024 - 3584 tiles per launch).  One workgroup per CU walks its XCD's share of the tile list; what changes against one workgroup per tile:
//   * the K-tile stream never drains: the half-tile fetches of K tiles t + 1 / t + 2 run on into the NEXT output tile (K tile 0 / 1 of
//     tile j + 1 are in flight while tile j's last K tile multiplies), so no tile but the first pays the ring's priming, and no workgroup
//     dispatch, address set-up or kernarg load stands between two tiles;
//   * the MFMAs are issued with the operands SWAPPED (weights as the A operand): lane (nl, kq) of a 16 x 16 result then holds four
//     consecutive COLUMNS n = 4 kq + {0..3} of ONE row m = nl, not a 4-row column strip -- a row scale per lane and tile, packed bf16 pairs
//     straight out of v_cvt_pk_bf16_f32, and two v_permlane32_swap / v_permlane16_swap pairs per two tiles leave every lane with 16
//     contiguous output bytes: the epilogue touches NO LDS (the old one transposed the tile through all 144 KiB of it, so nothing could be
//     fetched under it) and issues ~half the VALU instructions;
//   * row / column scales and the bias of tile j arrive by LDS-DMA during tile j's first K tile (parity-double-buffered 2.5 KiB slots behind
//     the K-tile buffers): no compiler-visible global load whose wait would be a vmcnt(0) on the running DMA queue;
//   * the first K tile of an output tile multiplies into C = 0 (a wave-uniform branch), so no accumulator is ever cleared by hand.
// The epilogue's 16 global stores per wave are younger than every fetch the next vmcnt(4) needs (loads retire in order among loads, so "at
// most 4 outstanding" still means all but the 4 youngest fetches have landed; the stores only make that wait stricter).
// Full tiles only (M, N multiples of 256), K >= 256, scale pointers 16-byte aligned: everything else keeps gemm8_p8_kernel.
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int kPScale = 2 * kBuf;             // the scale slots sit behind the two K-tile buffers
constexpr int kPSlot = 2560;                  // row scales 1 KiB | column scales 1 KiB | bias 512 B
constexpr int kPSmem = kPScale + 2 * kPSlot;  // 136192 B

// a, b: one dword per lane of two 16 x 16 result tiles in the swapped-operand layout (lane row q = lane >> 4 holds columns 4 q .. 4 q + 3).
// Afterwards a / b hold what lane rows (0, 1) resp. (2, 3) of tile a held, if this lane's row is 0 or 1, and the same of tile b otherwise:
// lane row q then owns columns 8 q .. 8 q + 7 of the 32 columns of the pair (a = the first four of them, b = the second four).
__device__ __forceinline__ void lane_rows_pair_up(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);        // a: [a0 a1 b0 b1], b: [a2 a3 b2 b3]   (lane rows of a | b)
  const auto s = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);  // a: [a0 a2 b0 b2], b: [a1 a3 b1 b3]
  a = s[0];
  b = s[1];
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm8_p8p_kernel(P8Args p) {
  constexpr bool IS_INT = (EPI == P8_INT8_SCALED || EPI == P8_INT32);
  constexpr bool SCALED = (EPI == P8_INT8_SCALED || EPI == P8_FP8_ROWWISE);
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nl = lane & 15, kq = lane >> 4;

  // this workgroup's tiles: XCD x (= id % 8) owns the contiguous range [x T / 8, (x + 1) T / 8) of the tile order (groups of GR tile rows
  // walking N together, as in gemm8_p8_kernel), its workgroups walk it side by side
  const int per = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int T = p.tiles_m * p.tiles_n;
  const int lo = (int)(((int64_t)xcd * T) >> 3), hi = (int)(((int64_t)(xcd + 1) * T) >> 3);
  if (hi - lo <= slot) return;
  const int mine = (hi - lo - slot + per - 1) / per;
  const int ktiles = p.K >> 7;
  const int GR = p.group_rows, group = GR * p.tiles_n;
  auto origin = [&](int j, int& m0, int& n0) {
    const int id = lo + slot + j * per;
    const int g0 = (id / group) * GR;
    const int gsz = min(GR, p.tiles_m - g0);
    m0 = (g0 + (id % group) % gsz) * 256;
    n0 = ((id % group) / gsz) * 256;
  };

  // DMA sources (full tiles: the same offsets for A and B, both have row pitch K): half-tile rows 16 w + 8 i + (lane >> 3)
  uint32_t off[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 16 * wave + 8 * i + (lane >> 3);
      off[h][i] = (uint32_t)(h * 128 + row) * (uint32_t)p.K + ((uint32_t)(((lane & 7) ^ (row >> 1)) & 7) << 4);
    }
  const uint32_t lds0 = lds_offset(smem);
  auto issue = [&](const uint8_t* asrc, const uint8_t* bsrc, int buf, int which) {  // which: 0 A-lo, 1 B-hi, 2 B-lo, 3 A-hi
    const bool is_a = (which == 0 || which == 3);
    const int half = (which == 1 || which == 3) ? 1 : 0;
    const uint32_t dst = lds0 + buf * kBuf + ((is_a ? 0 : 2) + half) * kHalf + wave * 2048;
    const uint8_t* src = is_a ? asrc : bsrc;
    dma_b128_s(src, off[half][0], dst);
    dma_b128_s(src, off[half][1], dst + 1024);
  };
  // the epilogue operands of a tile: waves 0 / 1 fetch the 256 row / column scales, waves 2 / 3 the two halves of the bias
  const uint32_t lane16 = (uint32_t)lane << 4, lane4 = (uint32_t)lane << 2;
  auto issue_scales = [&](int m0, int n0, int par) {
    if constexpr (SCALED) {
      const uint32_t dst = lds0 + kPScale + par * kPSlot;
      if (wave == 0) dma_b128_s(p.row_scale + m0, lane16, dst);
      if (wave == 1) dma_b128_s(p.col_scale + n0, lane16, dst + 1024);
      if (p.bias != nullptr && (wave == 2 || wave == 3)) dma_b32_s(p.bias + n0 + (wave - 2) * 128, lane4, dst + 2048 + (wave - 2) * 256);
    }
  };

  const int pos_lo = ((kq ^ (nl >> 1)) & 7) << 4;
  const int a_frag = wr * kHalf + nl * 128 + pos_lo;
  const int b_frag = (2 + (wc >> 1)) * kHalf + ((wc & 1) * 64 + nl) * 128 + pos_lo;

  f32x4 acc[8][4];
  u32x4 af[4][2], bf[4][2];
  auto load_a = [&](const char* buf, int mi) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int o = a_frag + (mi * 64 + mt * 16) * 128;
      af[mt][0] = *reinterpret_cast<const u32x4*>(buf + o);
      af[mt][1] = *reinterpret_cast<const u32x4*>(buf + (o ^ 64));
    }
  };
  auto load_b = [&](const char* buf) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int o = b_frag + (nt * 16) * 128;
      bf[nt][0] = *reinterpret_cast<const u32x4*>(buf + o);
      bf[nt][1] = *reinterpret_cast<const u32x4*>(buf + (o ^ 64));
    }
  };
  const int unit_scale = 127;  // E8M0 1.0
  // operands swapped (weights first): D[i = 4 kq + r][j = nl] = column n = 16 nt + i of row m = 16 mt + j
  auto multiply = [&](int mi, int nj, bool first) {
    if (first) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int nt = nj * 2 + n2;
          f32x4& c = acc[mi * 4 + mt][nt];
          if constexpr (IS_INT) {
            c = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bf[nt][0]), __builtin_bit_cast(i32x4, af[mt][0]),
                                                                               i32x4{0, 0, 0, 0}, 0, 0, 0));
          } else {
            const i32x8 av = {(int)af[mt][0].x, (int)af[mt][0].y, (int)af[mt][0].z, (int)af[mt][0].w,
                              (int)af[mt][1].x, (int)af[mt][1].y, (int)af[mt][1].z, (int)af[mt][1].w};
            const i32x8 bv = {(int)bf[nt][0].x, (int)bf[nt][0].y, (int)bf[nt][0].z, (int)bf[nt][0].w,
                              (int)bf[nt][1].x, (int)bf[nt][1].y, (int)bf[nt][1].z, (int)bf[nt][1].w};
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]" : "=&v"(c) : "v"(bv), "v"(av), "v"(unit_scale));
          }
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int nt = nj * 2 + n2;
          f32x4& c = acc[mi * 4 + mt][nt];
          if constexpr (IS_INT) {
            c = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bf[nt][0]), __builtin_bit_cast(i32x4, af[mt][0]),
                                                                               __builtin_bit_cast(i32x4, c), 0, 0, 0));
          } else {
            const i32x8 av = {(int)af[mt][0].x, (int)af[mt][0].y, (int)af[mt][0].z, (int)af[mt][0].w,
                              (int)af[mt][1].x, (int)af[mt][1].y, (int)af[mt][1].z, (int)af[mt][1].w};
            const i32x8 bv = {(int)bf[nt][0].x, (int)bf[nt][0].y, (int)bf[nt][0].z, (int)bf[nt][0].w,
                              (int)bf[nt][1].x, (int)bf[nt][1].y, (int)bf[nt][1].z, (int)bf[nt][1].w};
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(bv), "v"(av), "v"(unit_scale));
          }
        }
    }
    if constexpr (IS_INT) {  // the second k half of every tile in a second sweep (back-to-back MFMAs on one accumulator would wait for each other)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int nt = nj * 2 + n2;
          f32x4& c = acc[mi * 4 + mt][nt];
          c = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bf[nt][1]), __builtin_bit_cast(i32x4, af[mt][1]),
                                                                             __builtin_bit_cast(i32x4, c), 0, 0, 0));
        }
    }
  };
  auto seam = [&] {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  int m0, n0, m1 = 0, n1 = 0;
  origin(0, m0, n0);
  if (mine > 1) origin(1, m1, n1);
  const uint8_t* acur = p.a + (size_t)m0 * p.K;
  const uint8_t* bcur = p.b + (size_t)n0 * p.K;
  const uint8_t* anext = p.a + (size_t)m1 * p.K;
  const uint8_t* bnext = p.b + (size_t)n1 * p.K;

  // prologue: K tile 0 entirely and the B halves of K tile 1 (as if issued in q2 / q3 of a tile -1); wait for tile 0
  issue(acur, bcur, 0, 0); issue(acur, bcur, 0, 3); issue(acur, bcur, 0, 2); issue(acur, bcur, 0, 1);
  issue(acur + 128, bcur + 128, 1, 2); issue(acur + 128, bcur + 128, 1, 1);
  wait_vmcnt<4>();
  asm volatile("s_barrier" ::: "memory");
  if (wr == 1) asm volatile("s_barrier" ::: "memory");  // the stagger: wave row 1 runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  int gpar = 0;  // parity of the stream position (j ktiles + t): which buffer holds it
  for (int j = 0; j < mine; ++j) {
    const bool has_next = j + 1 < mine;
    for (int t = 0; t < ktiles; ++t, gpar ^= 1) {
      const char* buf = smem + gpar * kBuf;
      const bool first = (t == 0);
      // the sources of stream positions + 1 and + 2: this tile's next K tiles, or the next output tile's first ones
      const bool in1 = t + 1 < ktiles, in2 = t + 2 < ktiles;
      const bool ok1 = in1 || has_next, ok2 = in2 || has_next;
      const uint8_t* a1 = in1 ? acur + (size_t)(t + 1) * 128 : anext + (size_t)(t + 1 - ktiles) * 128;
      const uint8_t* b1 = in1 ? bcur + (size_t)(t + 1) * 128 : bnext + (size_t)(t + 1 - ktiles) * 128;
      const uint8_t* a2 = in2 ? acur + (size_t)(t + 2) * 128 : anext + (size_t)(t + 2 - ktiles) * 128;
      const uint8_t* b2 = in2 ? bcur + (size_t)(t + 2) * 128 : bnext + (size_t)(t + 2 - ktiles) * 128;
      // ---- q0: m-lo x n-lo ------------------------------------------------------------------------------------------------------------
      load_b(buf); __builtin_amdgcn_sched_barrier(0); load_a(buf, 0);
      if (first) issue_scales(m0, n0, j & 1);  // older than the fetches below: the q3 wait of this K tile covers them
      if (ok1) issue(a1, b1, gpar ^ 1, 0);
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(0, 0, first); __builtin_amdgcn_s_setprio(0);
      seam();
      // ---- q1: m-lo x n-hi ------------------------------------------------------------------------------------------------------------
      if (ok1) issue(a1, b1, gpar ^ 1, 3);
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(0, 1, first); __builtin_amdgcn_s_setprio(0);
      seam();
      // ---- q2: m-hi x n-hi ------------------------------------------------------------------------------------------------------------
      load_a(buf, 1);
      if (ok2) issue(a2, b2, gpar, 2);
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(1, 1, first); __builtin_amdgcn_s_setprio(0);
      seam();
      // ---- q3: m-hi x n-lo; stream position + 1 must have landed before the next phase reads it --------------------------------------------
      if (ok2) { issue(a2, b2, gpar, 1); wait_vmcnt<4>(); } else { wait_vmcnt<0>(); }
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(1, 0, first); __builtin_amdgcn_s_setprio(0);
      seam();
    }
    if (!has_next && wr == 0) asm volatile("s_barrier" ::: "memory");  // the last tile: wave row 1's final barrier must not wait for this row's epilogue
    if constexpr (!IS_INT) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // asm MFMA results -> VALU readers (see gemm8_p8_kernel)
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue of tile j (no LDS but the scale slot, no barrier: the other wave row multiplies on) -------------------------------------------
    {
      const int rbase = m0 + wr * 128, cbase = n0 + wc * 64;
      if constexpr (!SCALED) {
        uint32_t* out = reinterpret_cast<uint32_t*>(p.out);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<u32x4*>(out + (size_t)(rbase + mt * 16 + nl) * p.N + cbase + nt * 16 + kq * 4) = __builtin_bit_cast(u32x4, acc[mt][nt]);
      } else {
        const char* sl = smem + kPScale + (j & 1) * kPSlot;
        uint16_t* out = reinterpret_cast<uint16_t*>(p.out);
        auto scaled = [&](auto with_bias) {
          constexpr bool BIAS = decltype(with_bias)::value;
          // column pair by column pair (two 16-column tiles = the 32 columns one store covers): 8 + 8 scale / bias registers live at a time
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            f32x4 cs[2], bs[2];
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
              const int nt = 2 * pr + n2;
              cs[n2] = *reinterpret_cast<const f32x4*>(sl + 1024 + (wc * 64 + nt * 16 + kq * 4) * 4);
              if constexpr (BIAS) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(sl + 2048 + (wc * 64 + nt * 16 + kq * 4) * 2);
                bs[n2] = f32x4{bf16_lo_to_f32(b.x), bf16_hi_to_f32(b.x), bf16_lo_to_f32(b.y), bf16_hi_to_f32(b.y)};
              }
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
              const float rs = *reinterpret_cast<const float*>(sl + (wr * 128 + mt * 16 + nl) * 4);
              uint32_t d[2][2];
#pragma unroll
              for (int n2 = 0; n2 < 2; ++n2) {
                const int nt = 2 * pr + n2;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  if constexpr (EPI == P8_INT8_SCALED) {
                    // t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))   (int8_tensor.py:315-359)
                    v[r] = round_bf16((float)__builtin_bit_cast(i32x4, acc[mt][nt])[r] * rs) * cs[n2][r];
                  } else {
                    v[r] = acc[mt][nt][r] * rs * cs[n2][r];
                  }
                  if constexpr (BIAS) v[r] += bs[n2][r];
                }
                d[n2][0] = pack_bf16x2(v[0], v[1]);
                d[n2][1] = pack_bf16x2(v[2], v[3]);
              }
              lane_rows_pair_up(d[0][0], d[1][0]);
              lane_rows_pair_up(d[0][1], d[1][1]);
              const u32x4 o = {d[0][0], d[0][1], d[1][0], d[1][1]};
              *reinterpret_cast<u32x4*>(out + (size_t)(rbase + mt * 16 + nl) * p.N + cbase + pr * 32 + kq * 8) = o;
            }
          }
        };
        if (p.bias != nullptr) scaled(std::true_type{}); else scaled(std::false_type{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    m0 = m1; n0 = n1; acur = anext; bcur = bnext;
    if (j + 2 < mine) {
      origin(j + 2, m1, n1);
      anext = p.a + (size_t)m1 * p.K;
      bnext = p.b + (size_t)n1 * p.K;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// gemm8_p8h_kernel (round 5): the same phase-interleaved scheme on a 256 (M) x 128 (N) tile, for problems whose 256 x 256 tiles would leave
// half the chip idle (M <= 1024 on the TP shards: 28 .. 128 such tiles) while 256 x 128 tiles fill one round of it.
//   * 8 waves = 4 (m) x 2 (n): wave 4 g + s owns rows 64 s .. 64 s + 63 and columns 64 g .. 64 g + 63 (4 x 4 MFMA tiles, 64 accumulator
//     VGPRs).  A 64 x 64 wave tile reads (64 + 64) x 128 B of fragments per K tile -- 16 KiB, against 20 KiB for the 128 x 32 wave tile the
//     weight-streaming kernel uses at this workgroup tile: the loop is LDS-bound (176 KiB of reads + DMA writes per K tile at 128 B / clock
//     = 1375 cycles for 1024 cycles of MFMA per SIMD), so the wave shape that reads least wins.
//   * a K tile is TWO phases of 8 fp8 (16 int8) MFMAs: n-lo (all A fragments + the two n-lo B fragments read: 12 ds_read_b128), n-hi (4
//     reads).  Wave group g = 1 runs one barrier behind g = 0, as the two wave rows of the 256 x 256 kernel do: every SIMD holds one
//     wave of each group, one multiplies while the other reads and issues.
//   * three K tiles of [A-lo | A-hi | B] half tiles resident (144 KiB): tile t + 2 is fetched during tile t into the buffer tile t - 1
//     left -- every wave finished reading that one a full K tile earlier, whatever the stagger -- 4 DMA instructions in the first phase, 2 in
//     the second, then vmcnt(6): tile t + 1 has landed, tile t + 2's six stay in flight.
// Everything else (source swizzle, asm-pinned fp8 MFMAs, LDS-transposed epilogue, workgroup -> tile order) as in the kernel above.
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int kHBuf = 3 * kHalf;                    // one K tile: A-lo, A-hi, B
constexpr int kHSmem = 3 * kHBuf;                   // 147456 B (three K tiles; the epilogue's 8 x 64 x 144 B fit inside)

template <int EPI, int FORM = 0>
__global__ __launch_bounds__(512) void gemm8_p8h_kernel(P8Args p) {
  constexpr bool IS_INT = (EPI == P8_INT8_SCALED || EPI == P8_INT32);
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, s = wave & 3;
  const int nl = lane & 15, kq = lane >> 4;

  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const int S = p.split;  // K parts: consecutive work items (see gemm8_p8_kernel)
  const int ks = wg % S;
  wg /= S;
  const int GR = p.group_rows;
  const int group = GR * p.tiles_n;
  const int g0 = (wg / group) * GR;
  const int gsz = min(GR, p.tiles_m - g0);
  const int tm = g0 + (wg % group) % gsz, tn = (wg % group) / gsz;
  const int m0 = tm * 256, n0 = tn * 128;
  const int ktiles_all = p.K >> 7;
  const int kb = (int)((int64_t)ks * ktiles_all / S);
  const int ktiles = (int)((int64_t)(ks + 1) * ktiles_all / S) - kb;

  // DMA sources: half-tile rows 16 w + 8 i + (lane >> 3), chunk position lane & 7 holds global chunk (lane & 7) ^ ((row >> 1) & 7)
  uint32_t aoff[2][2], boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * wave + 8 * i + (lane >> 3);
    const uint32_t chunk = (uint32_t)(((lane & 7) ^ (row >> 1)) & 7) << 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) aoff[h][i] = (uint32_t)(min(m0 + h * 128 + row, p.M - 1) - m0) * (uint32_t)p.K + chunk;
    boff[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)p.K + chunk;
  }
  const char* abase = reinterpret_cast<const char*>(p.a) + (size_t)m0 * p.K + (size_t)kb * 128;
  const char* bbase = reinterpret_cast<const char*>(p.b) + (size_t)n0 * p.K + (size_t)kb * 128;
  const uint32_t lds0 = lds_offset(smem);
  auto issue = [&](int tile, int which) {  // which: 0 A-lo, 1 A-hi, 2 B
    if (tile >= ktiles) return;
    const uint32_t dst = lds0 + (tile % 3) * kHBuf + which * kHalf + wave * 2048;
    if (which == 2) dma_b128_x2(bbase + (size_t)tile * 128, boff[0], boff[1], dst);
    else dma_b128_x2(abase + (size_t)tile * 128, aoff[which][0], aoff[which][1], dst);
  };

  const int pos_lo = ((kq ^ (nl >> 1)) & 7) << 4;
  const int a_frag = (s >> 1) * kHalf + ((s & 1) * 64 + nl) * 128 + pos_lo;
  const int b_frag = 2 * kHalf + (g * 64 + nl) * 128 + pos_lo;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 af[4][2], bf[4][2];

  auto load_a = [&](const char* buf) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int off = a_frag + mt * 16 * 128;
      af[mt][0] = *reinterpret_cast<const u32x4*>(buf + off);
      af[mt][1] = *reinterpret_cast<const u32x4*>(buf + (off ^ 64));
    }
  };
  auto load_b = [&](const char* buf, int nj) {
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      const int nt = nj * 2 + n2;
      const int off = b_frag + nt * 16 * 128;
      bf[nt][0] = *reinterpret_cast<const u32x4*>(buf + off);
      bf[nt][1] = *reinterpret_cast<const u32x4*>(buf + (off ^ 64));
    }
  };
  const int unit_scale = 127;  // E8M0 1.0
  auto multiply = [&](int nj) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int nt = nj * 2 + n2;
        f32x4& c = acc[mt][nt];
        if constexpr (IS_INT) {
          c = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af[mt][0]), __builtin_bit_cast(i32x4, bf[nt][0]),
                                                                             __builtin_bit_cast(i32x4, c), 0, 0, 0));
        } else {
          const i32x8 av = {(int)af[mt][0].x, (int)af[mt][0].y, (int)af[mt][0].z, (int)af[mt][0].w,
                            (int)af[mt][1].x, (int)af[mt][1].y, (int)af[mt][1].z, (int)af[mt][1].w};
          const i32x8 bv = {(int)bf[nt][0].x, (int)bf[nt][0].y, (int)bf[nt][0].z, (int)bf[nt][0].w,
                            (int)bf[nt][1].x, (int)bf[nt][1].y, (int)bf[nt][1].z, (int)bf[nt][1].w};
          // volatile asm keeps the MFMAs inside their phase (see gemm8_p8_kernel)
          asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(av), "v"(bv), "v"(unit_scale));
        }
      }
    if constexpr (IS_INT) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int nt = nj * 2 + n2;
          f32x4& c = acc[mt][nt];
          c = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af[mt][1]), __builtin_bit_cast(i32x4, bf[nt][1]),
                                                                             __builtin_bit_cast(i32x4, c), 0, 0, 0));
        }
    }
  };
  auto seam = [&] {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // The epilogue's scales and bias are requested FIRST (VMEM returns in order: the counted waits below stay right, these being older than every
  // fetch) and kept as raw bits behind empty asm statements until the epilogue, so that no conversion -- and with it a vmcnt(0) -- is hoisted
  // into the loop.  On K = 1024 (8 K tiles) the epilogue's dependent scale loads were ~1 us of a 15 us launch.
  constexpr bool SCALED = (EPI == P8_INT8_SCALED || EPI == P8_FP8_ROWWISE);
  const int rbase = m0 + s * 64, cbase = n0 + g * 64;
  uint32_t pre_cs[4], pre_bias[4], pre_rs[4][4];
  if constexpr (SCALED) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = min(cbase + nt * 16 + nl, p.N - 1);
      pre_cs[nt] = __builtin_bit_cast(uint32_t, p.col_scale[n]);
      pre_bias[nt] = p.bias != nullptr ? (uint32_t)p.bias[n] : 0u;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pre_rs[mt][r] = __builtin_bit_cast(uint32_t, p.row_scale[min(rbase + mt * 16 + kq * 4 + r, p.M - 1)]);
  }
  // prologue: K tiles 0 and 1 entirely; wait for tile 0
  issue(0, 0); issue(0, 1); issue(0, 2); issue(1, 0); issue(1, 1); issue(1, 2);
  if (ktiles > 1) wait_vmcnt<6>(); else wait_vmcnt<0>();
  asm volatile("s_barrier" ::: "memory");
  if (g == 1) asm volatile("s_barrier" ::: "memory");  // the stagger: wave group 1 runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  if constexpr (FORM == 2) {  // (lab form) B-lo of tile 0 in registers before the loop
    load_b(smem, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  for (int t = 0; t < ktiles; ++t) {
    const char* buf = smem + (t % 3) * kHBuf;
    if constexpr (FORM == 0) {
      // ---- phase 0: all rows x n-lo; reads every A fragment and the n-lo B fragments; fetches the A halves of tile t + 2 ------------------
      load_b(buf, 0); __builtin_amdgcn_sched_barrier(0); load_a(buf);
      issue(t + 2, 0); issue(t + 2, 1);
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(0); __builtin_amdgcn_s_setprio(0);
      seam();
      // ---- phase 1: all rows x n-hi; fetches B of tile t + 2; tile t + 1 must have landed before the next phase reads it --------------------
      load_b(buf, 1);
      issue(t + 2, 2);
      if (t + 2 < ktiles) wait_vmcnt<6>(); else wait_vmcnt<0>();
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(1); __builtin_amdgcn_s_setprio(0);
      seam();
    } else if constexpr (FORM == 1) {
      // (lab form) all six fetches of tile t + 2 in phase 0: B gets 1.75 K tiles of flight instead of 1.0 -- 3 - 9 % slower
      load_b(buf, 0); __builtin_amdgcn_sched_barrier(0); load_a(buf);
      issue(t + 2, 0); issue(t + 2, 1); issue(t + 2, 2);
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(0); __builtin_amdgcn_s_setprio(0);
      seam();
      load_b(buf, 1);
      if (t + 2 < ktiles) wait_vmcnt<6>(); else wait_vmcnt<0>();
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(1); __builtin_amdgcn_s_setprio(0);
      seam();
    } else if constexpr (FORM == 2) {
      // (lab form) 8 + 8 fragment reads: phase 0 reads A (t), phase 1 reads B-hi (t) and B-lo (t + 1); tile t + 1 landed by the end of phase 0
      // -- 5 - 15 % slower
      load_a(buf);
      issue(t + 2, 0); issue(t + 2, 1); issue(t + 2, 2);
      if (t + 2 < ktiles) wait_vmcnt<6>(); else wait_vmcnt<0>();
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(0); __builtin_amdgcn_s_setprio(0);
      seam();
      load_b(buf, 1);
      if (t + 1 < ktiles) load_b(smem + ((t + 1) % 3) * kHBuf, 0);
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(1); __builtin_amdgcn_s_setprio(0);
      seam();
    } else {
      // (lab form) 12 reads + one fetch in phase 0, 4 reads + two fetches in phase 1: level (+- 2 %)
      load_b(buf, 0); __builtin_amdgcn_sched_barrier(0); load_a(buf);
      issue(t + 2, 0);
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(0); __builtin_amdgcn_s_setprio(0);
      seam();
      load_b(buf, 1);
      issue(t + 2, 1); issue(t + 2, 2);
      if (t + 2 < ktiles) wait_vmcnt<6>(); else wait_vmcnt<0>();
      seam();
      __builtin_amdgcn_s_setprio(1); multiply(1); __builtin_amdgcn_s_setprio(0);
      seam();
    }
  }
  if (g == 0) asm volatile("s_barrier" ::: "memory");  // group 0 catches up: from here on the LDS is free for every wave
  if constexpr (!IS_INT) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // asm MFMA results -> VALU / VMEM readers (see gemm8_p8_kernel)
  __builtin_amdgcn_sched_barrier(0);
  if (S > 1) {  // the K parts meet: 128 KiB per part, summed in part order by the last arriver, which goes on to the epilogue
    if (!split_k_meet2<16, 512, IS_INT, 2, 8>(reinterpret_cast<f32x4(&)[16]>(acc), p.ws, p.tickets, tm * p.tiles_n + tn, S, ks, tid,
                                                       reinterpret_cast<int*>(smem)))
      return;
  }

  // ---- epilogue: lane (col = nl, kq) holds rows 4 kq + {0..3} of each 16 x 16 tile ---------------------------------------------------------
  if constexpr (EPI == P8_INT32 || EPI == P8_FP8_RAW) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = rbase + mt * 16 + kq * 4 + r, n = cbase + nt * 16 + nl;
          if (m < p.M && n < p.N) reinterpret_cast<uint32_t*>(p.out)[(size_t)m * p.N + n] = __builtin_bit_cast(u32x4, acc[mt][nt])[r];
        }
  } else {
    float cs[4], bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      asm volatile("" : "+v"(pre_cs[nt]), "+v"(pre_bias[nt]));
      cs[nt] = __builtin_bit_cast(float, pre_cs[nt]);
      bias[nt] = bf16_lo_to_f32((uint16_t)pre_bias[nt]);
    }
    char* region = smem + wave * (64 * kEpiStride);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float rs[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        asm volatile("" : "+v"(pre_rs[mt][r]));
        rs[r] = __builtin_bit_cast(float, pre_rs[mt][r]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if constexpr (EPI == P8_INT8_SCALED) {
            v = round_bf16((float)__builtin_bit_cast(i32x4, acc[mt][nt])[r] * rs[r]) * cs[nt];  // (int8_tensor.py:315-359)
          } else {
            v = acc[mt][nt][r] * rs[r] * cs[nt];
          }
          if (p.bias != nullptr) v += bias[nt];
          *reinterpret_cast<uint16_t*>(region + (mt * 16 + kq * 4 + r) * kEpiStride + (nt * 16 + nl) * 2) = f32_to_bf16_bits(v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint16_t* out = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = i * 8 + (lane >> 3), piece = lane & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * kEpiStride + piece * 16);
      const int m = rbase + row, n = cbase + piece * 8;
      if (m < p.M && n + 8 <= p.N) *reinterpret_cast<u32x4*>(out + (size_t)m * p.N + n) = v;
    }
  }
}

// K parts of the 256 x 128 form (key 7 forces a count).  A part parks 128 KiB, half of the 256 x 256 kernel's, and on LONG K with few tiles
// that pays (profiles/p8h_split_sweep_r05.jsonl, fp8, cold; us, before -> with parts, hipBLASLt): K >= 8192, at least 512 rows, at most 128
// tiles, 2 - 4 parts (never more: 4 is the best count wherever more would fit) that make at least 128 workgroups --
//   qkv shard 1280 x 8192: M = 1024 29.5 -> 27.2 (27.8), M = 2048 45.2 -> 33.1 (42.3); down_proj 4096 x 14336: M = 512 48.0 -> 42.0 (49.4),
//   768 71.6 -> 58.1 (76.3), 1024 72.2 -> 62.5 (76.7); gate_up shard 7168 x 8192 at M = 512: 44.6 -> 42.6 (48.0).
// At K <= 4096 the parts lose in every cell (a part's loop is then shorter than its meeting), as do grids below 128 workgroups.
// Round 6, off the power-of-two grid (profiles/midm_offgrid_r06.jsonl, M = 192 .. 1536 x 9 forms x fp8 / int8): the row bound was 512 because the
// grid's next point down was 256 -- with two tile rows (257 .. 511 rows) the parts win as well: down_proj 4096 x 14336 at M = 320 / 384 / 448
// 44.1 / 45.9 / 46.6 (weight-streaming kernel) -> 39.2 / 39.1 / 40.4 with 4 parts, gate_up shard 42.1 / 42.8 / 43.0 -> 38.8 / 39.7 / 39.5 with 2
// (int8 the same); with ONE tile row (M = 192) 5 % ahead on the gate_up shard and 17 % behind on down_proj -> not taken.
int p8h_split_rule(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 255) / 256) * ((N + 127) / 128);
  if (K < 8192 || M <= 256 || tiles > 128) return 1;
  const int64_t S = std::min<int64_t>(4, 256 / tiles);
  return (S >= 2 && tiles * S >= 128) ? (int)S : 1;
}
int p8h_split(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 255) / 256) * ((N + 127) / 128), ktiles = K / 128;
  const int64_t fit = std::min<int64_t>({256 / tiles, ktiles, 16, (int64_t)(kSplitSlotFloats / ((size_t)tiles * 256 * 128)) * 4 / 5,
                                         (int64_t)kSplitMaxTickets / (tiles * 5)});
  if (g_p8_split > 0) return (int)std::max<int64_t>(1, std::min<int64_t>(g_p8_split, fit));
  return (int)std::max<int64_t>(1, std::min<int64_t>(p8h_split_rule(M, N, K), fit));
}

template <int EPI>
int launch_p8h(P8Args p, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 127) / 128;
  p.group_rows = (g_p8_group_rows >= 1 && g_p8_group_rows <= 64) ? g_p8_group_rows : 4;
  p.split = p8h_split(p.M, p.N, p.K);
  if (p.split > 1) {
    const int NG = (p.split + 3) / 4;
    if (int rc = splitk_workspace(stream, &p.ws, &p.tickets, (size_t)p.tiles_m * p.tiles_n * (p.split + NG) * 256 * 128, p.split)) return rc;
  }
#ifdef AO_LAB  // the measured-and-rejected loop forms (profiles/p8h_loop_forms_r05.jsonl) only exist in the laboratory build
  if (EPI == P8_FP8_ROWWISE && g_p8h_form == 1) {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_p8h_kernel<P8_FP8_ROWWISE, 1>), kHSmem, "hipFuncSetAttribute(gemm8_p8h_kernel)")) return rc;
    ao::launch(gemm8_p8h_kernel<P8_FP8_ROWWISE, 1>, dim3((unsigned)(p.tiles_m * p.tiles_n * p.split)), dim3(512), kHSmem, stream, p);
    return AO_OK;
  }
  if (EPI == P8_FP8_ROWWISE && g_p8h_form == 3) {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_p8h_kernel<P8_FP8_ROWWISE, 3>), kHSmem, "hipFuncSetAttribute(gemm8_p8h_kernel)")) return rc;
    ao::launch(gemm8_p8h_kernel<P8_FP8_ROWWISE, 3>, dim3((unsigned)(p.tiles_m * p.tiles_n * p.split)), dim3(512), kHSmem, stream, p);
    return AO_OK;
  }
  if (EPI == P8_FP8_ROWWISE && g_p8h_form == 2) {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_p8h_kernel<P8_FP8_ROWWISE, 2>), kHSmem, "hipFuncSetAttribute(gemm8_p8h_kernel)")) return rc;
    ao::launch(gemm8_p8h_kernel<P8_FP8_ROWWISE, 2>, dim3((unsigned)(p.tiles_m * p.tiles_n * p.split)), dim3(512), kHSmem, stream, p);
    return AO_OK;
  }
#endif
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_p8h_kernel<EPI>), kHSmem, "hipFuncSetAttribute(gemm8_p8h_kernel)")) return rc;
  ao::launch(gemm8_p8h_kernel<EPI>, dim3((unsigned)(p.tiles_m * p.tiles_n * p.split)), dim3(512), kHSmem, stream, p);
  AO_LAUNCH_CHECK("gemm8_p8h_kernel launch");
  return AO_OK;
}

thread_local int g_p8_persist = 0;  // (ao_gemm8_set_tuning key 6) 0 product rule, 1 never, 2 wherever the shape allows
constexpr int kP8ChipCUs = 256;     // MI355X: one persistent workgroup per CU, 32 per XCD

// The persistent form takes full tiles when there are more tiles than CUs, and any tile count at K < 4096, where its register-only epilogue and
// DMA-fetched scales are worth 2 - 7 % even with one tile per workgroup (profiles/p8_persist_ab_r06.jsonl, same process, alternating, cold weights:
// int8 at M = 16384 qkv 347 -> 331 us, o 222 -> 220, gate / up 784 -> 759, down 705 -> 707; fp8 K = 3584 / 4096 + 5 - 12 %, K = 1024 + 24 %;
// at <= 256 tiles and K >= 4096 the two are level, +- 1 %).  A loop form that dealt the 24 fragment reads of a K tile 8 / 4 / 8 / 4 over the
// four phases (the n-lo B fragments prefetched a K tile ahead) instead of 16 / 0 / 8 / 0 was 3 - 13 % SLOWER in every cell of that file
// ("balanced") and left the tree; so did a form that read the A fragments of the next phase under the wave's OWN MFMAs (a second fragment
// register set; no load phase but q0 then waits for an LDS round trip): int8 15 - 20 % slower, fp8 4 - 7 % (profiles/p8_persist_form1_ab_r06.jsonl) --
// the phase structure's premise, one wave row reads while the other multiplies, is where this loop is fastest.
bool p8_persistent_shape(int64_t M, int64_t N, int64_t K) {
  return M % 256 == 0 && N % 256 == 0 && K >= 256 && K % 128 == 0 && ((M / 256) * (N / 256) > kP8ChipCUs || K < 4096);
}
bool p8_persistent_takes(const P8Args& p) {
  if (g_p8_persist == 1) return false;
  const bool fits = p.M % 256 == 0 && p.N % 256 == 0 && p.K >= 256 && (reinterpret_cast<uintptr_t>(p.row_scale) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(p.col_scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
  return fits && (g_p8_persist == 2 || p8_persistent_shape(p.M, p.N, p.K));
}

template <int EPI>
int launch_p8(P8Args p, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  if (p8_persistent_takes(p)) {
    p.group_rows = (g_p8_group_rows >= 1 && g_p8_group_rows <= 64) ? g_p8_group_rows : 4;
    p.split = 1;
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    const unsigned grid = (unsigned)std::min<int64_t>(kP8ChipCUs, (tiles + 7) / 8 * 8);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_p8p_kernel<EPI>), kPSmem, "hipFuncSetAttribute(gemm8_p8p_kernel)")) return rc;
    ao::launch(gemm8_p8p_kernel<EPI>, dim3(grid), dim3(512), kPSmem, stream, p);
    AO_LAUNCH_CHECK("gemm8_p8p_kernel launch");
    return AO_OK;
  }
  // round 5 sweep (profiles/p8_group_rows_r05.jsonl): 4 tile rows per group measured 0 .. 7 % ahead of 8 on the Llama-3-8B int8 shapes at
  // M = 16384 (an XCD's L2 then holds 4 MB of A panels, its size, instead of 8 MB) and level elsewhere
  p.group_rows = (g_p8_group_rows >= 1 && g_p8_group_rows <= 64) ? g_p8_group_rows : 4;
  p.split = 1;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_p8_kernel<EPI>), kSmem, "hipFuncSetAttribute(gemm8_p8_kernel)")) return rc;
  ao::launch(gemm8_p8_kernel<EPI>, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), kSmem, stream, p);
  AO_LAUNCH_CHECK("gemm8_p8_kernel launch");
  return AO_OK;
}

}  // namespace

// epi: 0 int8 scaled (bf16 out), 1 int32 out, 2 fp8 rowwise (bf16 out), 3 fp8 raw fp32 out.  K % 128 == 0, N % 8 == 0,
// 256 * K < 4 GiB (32-bit in-tile offsets).
bool gemm8_p8_fits(int64_t M, int64_t N, int64_t K) { return K % 128 == 0 && N % 8 == 0 && 256 * K < (1ll << 32) && M > 0 && N > 0; }
// The shapes the 256 x 128 form takes (product rule, fp8 and int8): more than half a round and at most one round of the chip in 256 x 128 tiles,
// above 128 rows.  profiles/p8h_sweep_r05.jsonl (fp8, cold weights, 8 shapes x M = 256 .. 2048): inside the band it is ahead of every other
// kernel of the library in all 14 cells (1.1 - 1.5 x) and of hipBLASLt in 13 (o 8192 x 1024 at M = 1024: 0.98); at exactly 128 tiles it loses
// to the weight-streaming kernel in all 4 cells measured, below that by more.
// Round 6 (profiles/midm_offgrid_r06.jsonl): on that grid M was a multiple of 256, so the weight-streaming kernel's 128-row slabs always needed a
// second round of the chip inside the band.  With an odd count of 128-row slabs they may not (qkv 6144 x 4096 at M = 576 / 640: 3 x 48 = 144 tiles
// here, 5 x 48 = 240 slabs there): 29.5 / 29.3 us here against 24.7 / 24.5 streaming (hipBLASLt 26.2; int8 30.9 / 31.3 against 24.9 / 25.3) -- such
// shapes stay with the weight-streaming kernel (measured at K = 4096; shorter K not measured and left as it was).
bool gemm8_p8h_band(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 255) / 256) * ((N + 127) / 128), slabs = ((M + 127) / 128) * ((N + 127) / 128);
  if (M <= 128 || !gemm8_p8_fits(M, N, K)) return false;
  if (p8h_split_rule(M, N, K) > 1) return true;  // (the rule: with K parts, above)
  return tiles > 128 && tiles <= 256 && !(slabs <= 256 && K >= 4096);
}
int gemm8_p8h_parts(int64_t M, int64_t N, int64_t K) { return p8h_split_rule(M, N, K); }  // (product rule; host logic only)
void gemm8_p8_set_group_rows(int v) { g_p8_group_rows = v; }
void gemm8_p8_set_split(int v) { g_p8_split = v; }
void gemm8_p8_set_persistent(int v) { g_p8_persist = v; }
bool gemm8_p8_persistent_shape(int64_t M, int64_t N, int64_t K) { return p8_persistent_shape(M, N, K); }
void gemm8_p8h_set_form(int v) { g_p8h_form = v; }

// the 256 x 128 form (same epi numbering and shape limits)
int gemm8_p8h(int epi, const uint8_t* a, const uint8_t* b, const float* row_scale, const float* col_scale, const uint16_t* bias, void* out,
              int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  P8Args p{a, b, row_scale, col_scale, bias, out, (int)M, (int)N, (int)K, 0, 0, 0, 1, nullptr, nullptr};
  switch (epi) {
    case P8_INT8_SCALED: return launch_p8h<P8_INT8_SCALED>(p, stream);
    case P8_INT32: return launch_p8h<P8_INT32>(p, stream);
    case P8_FP8_ROWWISE: return launch_p8h<P8_FP8_ROWWISE>(p, stream);
    default: return launch_p8h<P8_FP8_RAW>(p, stream);
  }
}

int gemm8_p8(int epi, const uint8_t* a, const uint8_t* b, const float* row_scale, const float* col_scale, const uint16_t* bias, void* out,
             int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  P8Args p{a, b, row_scale, col_scale, bias, out, (int)M, (int)N, (int)K, 0, 0, 0, 1, nullptr, nullptr};
  switch (epi) {
    case P8_INT8_SCALED: return launch_p8<P8_INT8_SCALED>(p, stream);
    case P8_INT32: return launch_p8<P8_INT32>(p, stream);
    case P8_FP8_ROWWISE: return launch_p8<P8_FP8_ROWWISE>(p, stream);
    default: return launch_p8<P8_FP8_RAW>(p, stream);
  }
}

}  // namespace ao
