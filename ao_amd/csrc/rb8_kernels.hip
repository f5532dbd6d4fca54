// 1-byte-weight linears whose GEMM grid would not fill the chip, on gfx950: float8 rowwise at 32 < M (the TP-sharded
// Llama-70B shapes of BASELINE config 4 at batched-decode sizes) and int8 dynamic at any M with few output tiles
// (decode and small batches of config 3's model).  The op is then bound by streaming the weights once.
//
//   fp8 :  y[M,N] = bf16((a . b^T) * scale_a[m] * scale_b[n] + bias[n])     aten::_scaled_mm rowwise, float8/inference.py:104-123
//   int8:  y[M,N] = bf16(bf16(i32(a . b^T) * sx[m]) * sw[n] + bias[n])      _int_mm + scales, int8_tensor.py:305-359
//
// Same structure as the batched int4 kernel (int4_kernels.hip, int4_mm_rb_kernel) without the dequant: a wave owns
// one 16-wide n-tile and all 128 rows of the slab.  Its weight rows go HBM -> wave-private LDS ring (full 128-byte lines
// per request) -> registers as the B operand of v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) or of two
// v_mfma_i32_16x16x64_i8 (lane (n, kq) holds k = 16 kq .. +15 and 64 + 16 kq .. +15 of the 128-k step);
// the activation tile [128 rows][128 B] is staged once per workgroup by
// LDS-DMA with a source-side swizzle (chunk position c' of row r holds global 16-byte chunk c' ^ ((r >> 1) & 7)) so
// that the two ds_read_b128 of an A operand are bank-conflict free.  Activations are fetched 2 steps ahead (L2 hits),
// weights 5 steps ahead (with 2 the loop ran at 0.9 us per step: too few HBM bytes in flight); one hand-counted
// s_waitcnt vmcnt + one LDS-only barrier per step.  K is cut into parts when the grid is small; the
// parts meet through the fence-free split-K workspace (splitk.h), the last arriver applies the scales.
#include "common.h"
#include "lds_dma.h"
#include "quant_math.h"
#include "splitk.h"

#include <algorithm>
#include <type_traits>

namespace ao {
namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int kStages = 3;   // activation ring (shared), filled 2 steps ahead
// weight ring (per wave), filled kWStages - 1 steps ahead: the HBM stream needs the bytes in flight.  6 stages; 5 for the
// 8-wave MX form, whose scale rings would otherwise push the workgroup past 160 KiB of LDS
// (also 5 for the 64-row MX slab, which then fits two workgroups per CU)
constexpr int w_stages(int waves, int kind, int mt, bool slim = false) {
  return slim ? 3 : (kind == 2 && (waves == 8 || mt == 4)) ? 5 : 6;
}
// Activation ring: 3 stages (2 steps ahead).  Round 5 measured 5 / 4 stages for the rowwise kinds (the ring's depth looked like the
// loop's bound: 1100 cycles per step for every tile width, ~ half an LDS-DMA round trip): the step time did not move -- it is the sum
// of one wave's own barrier + issue + fragment-read + MFMA chain (timing probes of the traced build: 430 + 125 + 250 + 256 cycles at
// 32 columns), not a latency -- and the longer priming cost 3 % on the short-K shapes.  256-row slabs (mt = 16): a stage is 32 KiB.
constexpr int a_stages(int waves, int kind, int mt) { return kStages; }
// RB8_FP8_GROUPED: rowwise e4m3 like RB8_FP8, rows grouped by expert like RB8_MX (Float8Tensor's _grouped_mm, float8_tensor.py:1085-1122)
enum Rb8Kind { RB8_FP8 = 0, RB8_INT8 = 1, RB8_MX = 2, RB8_FP8_GROUPED = 3 };
constexpr bool GROUPED_KIND(int kind) { return kind == RB8_MX || kind == RB8_FP8_GROUPED; }

struct Rb8Args {
  const uint8_t* a;       // [M][K] e4m3 / int8
  const uint8_t* b;       // [N][K] e4m3 / int8; grouped kinds: [E][N][K]
  const float* scale_a;   // [M]           (rowwise kinds)
  const float* scale_b;   // [N]; RB8_FP8_GROUPED: [E][N]
  const uint16_t* bias;   // [N] bf16 or null
  uint16_t* y;            // [M][N] bf16
  int M, N, K;
  const uint8_t* a_mx;    // MX: [M][K/32] e8m0
  const uint8_t* b_mx;    // MX: [E][N][K/32] e8m0
  const int32_t* offs;    // MX: [E] cumulative group ends (null: one group)
  // MX stream-K, round 6: a SECOND weight tensor of the same shape multiplied by the same activations in the same launch (an MoE layer's
  // w1 and w3: x @ w1, x @ w3) -- its column tiles follow the first's in every slab; null: one product
  const uint8_t* b2;
  const uint8_t* b2_mx;
  uint16_t* y2;
  int slabs, E;           // MX: 128-row slabs per group the grid provides (grid.y = E * slabs)
  float* ws;
  unsigned* tickets;
  unsigned long long* trace;  // profiling build only
  int ablate;  // profiling build (TRACE) only: 1 no MFMAs, 2 no fragment reads, 4 no weight DMAs, 8 no activation DMAs, 16 DMA wait + barrier on even steps only -- wrong results, timing probes
};

// TRACE (profiling build): s_memtime stamps of wave 0, 16 u64 per workgroup: entry, ring primed, barrier of steps 0..7 passed,
// loop done, meeting done, exit
// MT = 16-row m-tiles per slab (8, 4, 2): small batches / token groups stage, read and multiply only the rows they can have.
// QS (MX): k steps per scale fetch -- 1: a dword DMA per row, step and operand; 4: one 16-byte DMA per row and FOUR steps, two slots
// per wave and operand (K % 512 == 0, no K split).  The dword DMAs cost a quarter of the decode kernel (mx_stream_kernel below).
// SM (round 4, rowwise kinds, 8 waves x 128 rows): the waves stand 2 (m halves) x 4 (pairs of n-tiles) instead of 1 x 8 -- wave (wm, wn)
// multiplies m-tiles 4 wm .. 4 wm + 3 by n-tiles 2 wn, 2 wn + 1.  Same DMAs, same rings, same LDS layout (wave w still FETCHES n-tile w
// and its eighth of the activation tile; the step's barrier already orders everybody's fetches before anybody's reads), same 8
// accumulators and MFMAs per wave and step -- but 4 + 2 operand fragments to read per step instead of 8 + 1: 12 ds_read_b128 for 18,
// on a loop that is bound by the LDS reads (144 KiB per step and workgroup at 128 B / clk against 512 cycles of MFMA per SIMD).
template <int WAVES, int KIND, int MT = 8, bool TRACE = false, bool SLIM = false, int QS = 1, bool SM = false>
__global__ __launch_bounds__(64 * WAVES) void rb8_kernel(Rb8Args p) {
  static_assert(!SM || (WAVES == 8 && MT == 8 && (KIND == RB8_FP8 || KIND == RB8_INT8)), "rb8_kernel: the 2 x 4 wave arrangement is built for 8 waves x 128 rows, rowwise kinds");
  constexpr int MH = MT / 2;  // SM: m-tiles per wave (wave (wm, wn): m-tiles MH wm .. + MH - 1, n-tiles 2 wn, 2 wn + 1; acc[2 i + j])
  constexpr int SCL = SLIM ? 64 : 256;  // bytes of one scale slot (one dword per row: 16 rows -> 64 B; unmasked DMAs write 256)
  static_assert(QS == 1 || (QS == 4 && KIND == RB8_MX && !SLIM), "rb8_kernel: 4-step scale fetches are an MX form");
  unsigned long long ts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [13] group found, [14] addresses ready
  if (TRACE) ts[0] = __builtin_amdgcn_s_memtime();
  constexpr bool INT8 = (KIND == RB8_INT8), MX = (KIND == RB8_MX), GROUPED = (KIND == RB8_MX || KIND == RB8_FP8_GROUPED);
  constexpr int ADMA = 2 * MT / WAVES;  // activation DMAs per wave and stage (8 rows each)
  static_assert(ADMA >= 1, "every wave issues the same number of DMAs per stage");
  constexpr int kABuf = MT * 2048;      // one activation stage: 16 MT rows x 128 k bytes
  constexpr int BM = 16 * MT;
  constexpr int RPW = BM / WAVES;   // MX: activation-scale rows fetched per wave
  constexpr int kWStages = w_stages(WAVES, KIND, MT, SLIM);
  constexpr int KA = a_stages(WAVES, KIND, MT);  // activation ring
  static_assert(!SLIM || (KIND == RB8_MX && BM / WAVES == 16), "SLIM: 16 activation-scale rows per wave");
  // [3][128][128 B] a | [WAVES][6][2 KiB] b | MX: [3][WAVES][256 B] a scales | [WAVES][6][256 B] b scales
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  const int ntiles = p.N >> 4;
  // (bx, by): the workgroup's column tile and slab; ks of S: its K part
  const int bx = blockIdx.x, by = blockIdx.y, ks = blockIdx.z, S = gridDim.z, gx = gridDim.x;
  const int tile = bx * WAVES + wave;
  const int tile_c = min(tile, ntiles - 1);  // tiles past N alias the last one; never stored
  const int ksteps = p.K >> 7;
  // (32-bit: ksteps < 2^24 and S <= 16 -- the 64-bit division this used to be was ~300 scalar instructions at the head of every workgroup)
  const int k0 = (int)(((unsigned)ksteps * (unsigned)ks) / (unsigned)S);
  const int nk = (int)(((unsigned)ksteps * (unsigned)(ks + 1)) / (unsigned)S) - k0;
  // Two more round-4 A/Bs on this kernel, both dropped (profiles/rb8_ab_r04.txt, cold 70B / TP8 shards at M = 128 / 64): the two weight
  // DMAs (and the two activation DMAs) of a stage behind ONE M0 write: +- 0.1 us here and on the MX stream-K kernel; the epilogue's
  // scales requested ahead of the meeting (at the head of the kernel, behind the loop, or behind the drain -- handed on through LDS):
  // + 2 us per launch in all three placements.
  // rows of this workgroup: [m0, m_end) -- a slab of the matrix, or of one expert's token group.  Grouped kinds: blockIdx.y
  // enumerates the NON-EMPTY slabs in (expert, slab) order -- the y-th one is found from the group ends on the device, so the
  // grid needs ceil(M / BM) + E rows at most (not E x the slabs of the largest possible group) and an expert without tokens
  // costs nothing.
  int m0 = by * BM, m_end = p.M, expert = 0;
  if constexpr (GROUPED) {
    if (p.offs != nullptr) {
      int y = by, found = -1, begin = 0, end = 0;
      for (int e0 = 0; e0 < p.E && found < 0; e0 += 64) {  // 64 experts per pass: lane e owns expert e0 + e
        const int e = e0 + lane;
        const int lo = (e > 0 && e < p.E) ? p.offs[e - 1] : 0;
        const int hi = (e < p.E) ? p.offs[e] : lo;
        const int ns = (hi - lo + BM - 1) / BM;  // slabs of this expert (0 when it has no tokens)
        int incl = ns;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d);
          if (lane >= d) incl += t;
        }
        const unsigned long long hit = __ballot(incl > y);
        if (hit != 0ull) {
          const int l = __builtin_ctzll(hit);
          found = e0 + l;
          begin = __builtin_amdgcn_readlane(lo, l);
          end = __builtin_amdgcn_readlane(hi, l);
          y -= __builtin_amdgcn_readlane(incl - ns, l);
        } else {
          y -= __builtin_amdgcn_readlane(incl, 63);
        }
      }
      if (found < 0) return;  // uniform: past the last non-empty slab (before any DMA or barrier)
      expert = found;
      m0 = begin + y * BM;
      m_end = end;
    } else if (m0 >= m_end) {
      return;
    }
  }
  // m-tiles this slab really has, rounded up to a power of two: the k loop below is specialised on it, so a 32-row group in a
  // 64- or 128-row slab reads and multiplies 32 rows (the DMA counts stay static for the hand-counted waits; rows past the
  // group re-read its last row)
  const int mt_have = (min(m_end - m0, BM) + 15) >> 4;
  if (TRACE) ts[13] = __builtin_amdgcn_s_memtime();

  // weight DMA i (0, 1) of a step fetches rows 8 i + (lane >> 3) of the n-tile as FULL 128-byte lines (chunk position lane & 7,
  // same swizzle as the activations); half-line requests -- one lane group per 64 bytes -- ran the stream at 3.7 TB/s
  uint32_t boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * i + (lane >> 3);
    boff[i] = (uint32_t)row * (uint32_t)p.K + ((((lane & 7) ^ (row >> 1)) & 7) << 4);
  }
  const uint8_t* brows = p.b + ((size_t)expert * p.N + (size_t)tile_c * 16) * p.K;
  const uint32_t a_lds = lds_offset(smem);
  const uint32_t w_lds = a_lds + KA * kABuf + wave * (kWStages * 2048);
  // MX block scales: one dword (4 e8m0 bytes = the 4 blocks of a 128-k step) per row and step.  Wave w fetches the dwords of
  // activation rows RPW w .. + RPW - 1 (lanes past RPW repeat them into slots nobody reads) and of its own 16 weight rows.
  const uint32_t kb32 = (uint32_t)(p.K >> 5);
  const uint32_t asoff = MX ? (uint32_t)min(m0 + RPW * wave + (lane % RPW), m_end - 1) * kb32 : 0u;
  const uint32_t bsoff = (uint32_t)nl * kb32;
  const uint8_t* bsrows = MX ? p.b_mx + ((size_t)expert * p.N + (size_t)tile_c * 16) * kb32 : nullptr;
  const uint32_t as_lds = a_lds + KA * kABuf + WAVES * (kWStages * 2048);
  constexpr int ASB = (QS == 1) ? SCL : RPW * 16, BSB = 256;  // QS == 4: slot bytes (activation rows of a wave, its 16 weight rows: 16 B each)
  const uint32_t bs_lds = (QS == 1) ? as_lds + KA * WAVES * SCL + wave * (kWStages * SCL) : as_lds + 2 * WAVES * ASB + wave * (2 * BSB);
  auto issue_s = [&](int slot, int k) {  // QS == 4: the scales of steps k .. k + 3 (k % 4 == 0); past the end: the last block again
    const int kk = k0 + min(k, nk - 4);
    if (lane < RPW) dma_b128_s(p.a_mx + (size_t)kk * 4, asoff, as_lds + (slot * WAVES + wave) * ASB);
    if (lane < 16) dma_b128_s(bsrows + (size_t)kk * 4, bsoff, bs_lds + slot * BSB);
  };
  auto issue_w = [&](int stage, int k) {
    const int kk = k0 + min(k, nk - 1);
    dma_b128_nt_s(brows + (size_t)kk * 128, boff[0], w_lds + stage * 2048);
    dma_b128_nt_s(brows + (size_t)kk * 128, boff[1], w_lds + stage * 2048 + 1024);
    if constexpr (MX && QS == 1) {
      if (!SLIM || lane < 16) dma_b32_s(bsrows + (size_t)kk * 4, bsoff, bs_lds + stage * SCL);
    }
  };

  // ROLES (round 5, rowwise kinds): the lower half of the waves fetches the activation tile, the upper half the weights (two n-tiles
  // each).  VMEM retires in order per wave: a wave that requests w(k + 5) and then a(k + 2) cannot see a(k + 2) land before w(k + 5) has,
  // so the weight ring's five stages of HBM latency cover were really two (the round-5 trace: 1100 ticks per step whatever the tile
  // width -- half an HBM round trip).  With the kinds apart, an activation wave waits for L2 hits only and a weight wave has its whole
  // ring in flight; the step's barrier publishes both, as before.  Same LDS layout, same MFMAs, same bits.
  constexpr bool ROLES = !GROUPED;
  const bool is_a = wave < WAVES / 2;
  const uint8_t* brows2[2] = {nullptr, nullptr};
  uint32_t w_lds2[2] = {0u, 0u};
  if constexpr (ROLES) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nt = max(2 * (wave - WAVES / 2) + j, 0);  // (activation waves: unused)
      brows2[j] = p.b + (size_t)min(bx * WAVES + nt, ntiles - 1) * 16 * p.K;
      w_lds2[j] = a_lds + KA * kABuf + nt * (kWStages * 2048);
    }
  }
  // the epilogue's scales, requested before anything else (one row scale, one column scale, one bias value per thread at most) and
  // handed on through LDS after the loop: by then they have long landed, and the tail has no memory round trip left but its stores
  float pre_sa = 0.f, pre_sb = 0.f;
  uint32_t pre_bias = 0u;  // bf16 bits: converted in the epilogue (a use here would make the compiler wait for the load at the kernel's head)
  if constexpr (!GROUPED) {
    // (every part: any of them may turn out to be the last arriver)
    if (tid < BM) pre_sa = p.scale_a[min(m0 + tid, p.M - 1)];
    if (tid < 16 * WAVES) {
      const int col = min(bx * (16 * WAVES) + tid, p.N - 1);
      pre_sb = p.scale_b[col];
      if (p.bias != nullptr) pre_bias = p.bias[col];
    }
  }

  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // A operand of lane (row r = nl, kq): chunks kq and 4 + kq of the row, at positions chunk ^ ((r >> 1) & 7)
  const int pa = nl * 128 + (((kq ^ (nl >> 1)) & 7) << 4);  // second half: ^ 64; + 2048 per m-tile

  // Issue order (6 stages) is w(0..2) | a(0) w(3) | a(1) w(4), then per step a(k+2) w(k+5): when step k starts, the youngest requests
  // are a(k+1), w(k+4) (one stage: LPSC) and w(k+3) (2 DMAs, 3 with MX scales); everything older -- a(k), w(k) .. w(k+2) -- has landed.
  auto k_loop = [&](auto mtc) {
  constexpr int MTC = decltype(mtc)::value;
  // activation DMAs per wave and stage for the m-tiles this slab really has (8 rows each; with fewer 8-row blocks than waves
  // the last waves re-fetch the group's last row into rows nobody reads): a 32-row group in a 64-row slab costs the workgroup
  // 4 activation DMAs per step, not 8 -- the activation tile is half of what a CU's texture path moves per step
  // (ROLES: the activation tile's 2 MTC DMAs are shared out among the WAVES / 2 activation waves)
  constexpr int AD = ROLES ? 4 * MTC / WAVES : (2 * MTC >= WAVES) ? 2 * MTC / WAVES : 1;
  static_assert(AD >= 1, "rb8_kernel: every fetching wave issues the same number of DMAs per stage");
  constexpr int LPSC = AD + 2 + ((MX && QS == 1) ? 2 : 0);
  uint32_t aoff[AD];
#pragma unroll
  for (int i = 0; i < AD; ++i) {
    const int row = 8 * (AD * wave + i) + (lane >> 3);  // (ROLES: wave < WAVES / 2 covers the tile; the weight waves never issue these)
    aoff[i] = (uint32_t)min(m0 + min(row, BM - 1), m_end - 1) * (uint32_t)p.K + ((((lane & 7) ^ (row >> 1)) & 7) << 4);
  }
  auto issue_a = [&](int stage, int k) {  // k clamped: the fills past the end re-read the last step (unused)
    const int kk = k0 + min(k, nk - 1);
    if (TRACE && (p.ablate & 8)) return;
#pragma unroll
    for (int i = 0; i < AD; ++i) dma_b128_s(p.a + (size_t)kk * 128, aoff[i], a_lds + stage * kABuf + (AD * wave + i) * 1024);
    if constexpr (MX && QS == 1) {
      if (!SLIM || lane < 16) dma_b32_s(p.a_mx + (size_t)kk * 4, asoff, as_lds + (stage * WAVES + wave) * SCL);
    }
  };
  auto issue_w2 = [&](int stage, int k) {  // ROLES: a weight wave's two n-tiles
    const int kk = k0 + min(k, nk - 1);
    if (TRACE && (p.ablate & 4)) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dma_b128_nt_s(brows2[j] + (size_t)kk * 128, boff[0], w_lds2[j] + stage * 2048);
      dma_b128_nt_s(brows2[j] + (size_t)kk * 128, boff[1], w_lds2[j] + stage * 2048 + 1024);
    }
  };
  if (TRACE) ts[14] = __builtin_amdgcn_s_memtime();
  if constexpr (QS == 4) issue_s(0, 0);  // then the block of steps 4 j + 4 .. at the head of step 4 j + 1 (older than a(4 j + 3): landed by then)
  if constexpr (ROLES) {
    if (is_a) {
#pragma unroll
      for (int i = 0; i < KA - 1; ++i) issue_a(i, i);
    } else {
#pragma unroll
      for (int i = 0; i < kWStages - 1; ++i) issue_w2(i, i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < kWStages - 3; ++i) issue_w(i, i);
    issue_a(0, 0); issue_w(kWStages - 3, kWStages - 3);
    issue_a(1, 1); issue_w(kWStages - 2, kWStages - 2);
  }
  if (TRACE) ts[1] = __builtin_amdgcn_s_memtime();
  int stage = 0, wstage = 0;
  if constexpr (ROLES) {
    // Round 5: the step's operand fragments are double-buffered in REGISTERS.  The round-5 trace showed 1000 - 1200 ticks per step for every
    // tile width, ring depth and fetch split -- the time of one wave's own instruction chain: the compiler emitted the step as four
    // "ds_read x 4-6, wait, MFMA x 2" groups, an LDS round trip in front of each (one or two waves per SIMD, all parked at the same
    // barrier: nothing else to issue).  Now step k + 1's barrier, its ring refills and ALL its fragment reads are issued before step
    // k's MFMAs, which run from registers read a step earlier: the LDS latency sits under 8 MFMAs (256 matrix cycles) instead of
    // between them.  Same products into the same accumulators in the same order: same bits.
    constexpr int NA = SM ? MH : MTC, NB = SM ? 2 : 1;
    struct Frags { u32x4 a0[NA], a1[NA], b0[NB], b1[NB]; };
    const int wm = wave & 1, wn = wave >> 1;  // (SM)
    int ksync = 0;  // the next step to synchronise
    auto sync_issue_read = [&](Frags& f) {
      // step ksync's operands have landed (an activation wave may have a(k + 1) .. a(k + KA - 2) in flight, a weight wave
      // w(k + 1) .. w(k + kWStages - 2), four DMAs a stage) and everyone has finished READING step ksync - 1 (lgkmcnt(0) before the
      // barrier): its slots are refilled, then this step's fragments are requested
      static_assert(AD * (KA - 2) <= 63 && 4 * (kWStages - 2) <= 63, "rb8_kernel: vmcnt is a 6-bit counter");
      // (timing probe 16 of the traced build: the DMA wait and the barrier on even steps only -- what a 256-byte K step would save at best)
      const bool skip_sync = TRACE && (p.ablate & 16) && (ksync & 1);
      if (!skip_sync) { if (is_a) wait_vmcnt<AD * (KA - 2)>(); else wait_vmcnt<4 * (kWStages - 2)>(); }
      // (the LDS wait as a BUILTIN, not inside the asm: the compiler's wait-count pass has to see it, or it takes the fragments read
      // a step ago for still in flight behind the new reads and waits for those in front of every MFMA -- lgkmcnt(13) .. (0))
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt, expcnt untouched
      if (!skip_sync) asm volatile("s_barrier" ::: "memory");
      if (TRACE && ksync < 8) ts[2 + ksync] = __builtin_amdgcn_s_memtime();
      if (is_a) issue_a((stage == 0) ? KA - 1 : stage - 1, ksync + KA - 1);
      else issue_w2((wstage == 0) ? kWStages - 1 : wstage - 1, ksync + kWStages - 1);
      const char* A = smem + stage * kABuf;
      if (TRACE && (p.ablate & 2)) {
      } else if constexpr (SM) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const char* Wj = smem + KA * kABuf + ((2 * wn + j) * kWStages + wstage) * 2048;
          f.b0[j] = *reinterpret_cast<const u32x4*>(Wj + pa);
          f.b1[j] = *reinterpret_cast<const u32x4*>(Wj + (pa ^ 64));
        }
#pragma unroll
        for (int i = 0; i < MH; ++i) {
          f.a0[i] = *reinterpret_cast<const u32x4*>(A + (MH * wm + i) * 2048 + pa);
          f.a1[i] = *reinterpret_cast<const u32x4*>(A + (MH * wm + i) * 2048 + (pa ^ 64));
        }
      } else {
        const char* W = smem + KA * kABuf + (wave * kWStages + wstage) * 2048;
        f.b0[0] = *reinterpret_cast<const u32x4*>(W + pa);
        f.b1[0] = *reinterpret_cast<const u32x4*>(W + (pa ^ 64));
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
          f.a0[mt] = *reinterpret_cast<const u32x4*>(A + mt * 2048 + pa);
          f.a1[mt] = *reinterpret_cast<const u32x4*>(A + mt * 2048 + (pa ^ 64));
        }
      }
      stage = (stage == KA - 1) ? 0 : stage + 1;
      wstage = (wstage == kWStages - 1) ? 0 : wstage + 1;
      ++ksync;
      __builtin_amdgcn_sched_barrier(0);  // the reads are ISSUED here, ahead of the MFMAs of the step before
    };
    auto one = [&](f32x4& c, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
      if constexpr (INT8) {  // acc holds int32 bit patterns
        i32x4 ci = __builtin_bit_cast(i32x4, c);
        ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), ci, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, b1), ci, 0, 0, 0);
        c = __builtin_bit_cast(f32x4, ci);
      } else {
        const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        const i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, c, 0, 0, 0, 127, 0, 127);
      }
    };
    auto mma = [&](const Frags& f) {
      if (TRACE && (p.ablate & 3)) {
      } else if constexpr (SM) {
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) one(acc[2 * i + j], f.a0[i], f.a1[i], f.b0[j], f.b1[j]);
      } else {
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) one(acc[mt], f.a0[mt], f.a1[mt], f.b0[0], f.b1[0]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    Frags f0, f1;
    sync_issue_read(f0);
    // (the explicit LDS waits on the paths WITHOUT new reads are for the compiler's wait-count pass: where paths merge it assumes the
    // worst of them, and would put lgkmcnt waits for the NEW reads in front of every MFMA of the old fragments)
    for (int k = 0; k < nk; k += 2) {
      if (k + 1 < nk) sync_issue_read(f1); else __builtin_amdgcn_s_waitcnt(0xC07F);
      mma(f0);
      if (k + 1 < nk) {
        if (k + 2 < nk) sync_issue_read(f0); else __builtin_amdgcn_s_waitcnt(0xC07F);
        mma(f1);
      }
    }
    return;
  }
  for (int k = 0; k < nk; ++k) {
    // (3 weight stages: w(k) is issued right behind a(k), so only a(k + 1) and w(k + 1) -- one stage -- may still be in flight)
    // (QS == 4: at k % 4 == 2 the two scale requests of step k - 1 are younger than a(k) too)
    if constexpr (QS == 4) { if ((k & 3) == 2) wait_vmcnt<LPSC + 2 + 2>(); else wait_vmcnt<LPSC + 2>(); }
    else if constexpr (kWStages >= 4) wait_vmcnt<LPSC + 2 + (MX ? 1 : 0)>(); else wait_vmcnt<LPSC>();
    // everyone's share of the activation tile has landed, and everyone has finished reading step k - 1
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (TRACE && k < 8) ts[2 + k] = __builtin_amdgcn_s_memtime();
    if constexpr (QS == 4) {
      if ((k & 3) == 1) issue_s(((k >> 2) + 1) & 1, (k & ~3) + 4);
    }
    issue_a((stage == 0) ? KA - 1 : stage - 1, k + KA - 1);
    issue_w((wstage == 0) ? kWStages - 1 : wstage - 1, k + kWStages - 1);
    const char* A = smem + stage * kABuf;
    if constexpr (SM) {
      // wave (wm, wn): n-tiles 2 wn + j (fetched by waves 2 wn + j), m-tiles 4 wm + i; acc[2 i + j]
      const int wm = wave & 1, wn = wave >> 1;
      u32x4 b0[2], b1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* Wj = smem + KA * kABuf + ((2 * wn + j) * kWStages + wstage) * 2048;
        b0[j] = *reinterpret_cast<const u32x4*>(Wj + pa);
        b1[j] = *reinterpret_cast<const u32x4*>(Wj + (pa ^ 64));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(A + (4 * wm + i) * 2048 + pa);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(A + (4 * wm + i) * 2048 + (pa ^ 64));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (INT8) {
            i32x4 c = __builtin_bit_cast(i32x4, acc[2 * i + j]);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0[j]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, b1[j]), c, 0, 0, 0);
            acc[2 * i + j] = __builtin_bit_cast(f32x4, c);
          } else {
            const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
            const i32x8 bfj = {(int)b0[j].x, (int)b0[j].y, (int)b0[j].z, (int)b0[j].w, (int)b1[j].x, (int)b1[j].y, (int)b1[j].z, (int)b1[j].w};
            acc[2 * i + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bfj, acc[2 * i + j], 0, 0, 0, 127, 0, 127);
          }
        }
      }
      stage = (stage == KA - 1) ? 0 : stage + 1;
      wstage = (wstage == kWStages - 1) ? 0 : wstage + 1;
      continue;
    }
    const char* W = smem + KA * kABuf + (wave * kWStages + wstage) * 2048;
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(W + pa);  // the n-tile's 16 rows are laid out like an m-tile
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(W + (pa ^ 64));
    const i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    // MX: the scale byte of lane group kq is that of 32-k block kq of the step (operand layout probed on gfx950, stream8_kernels.hip)
    [[maybe_unused]] const char* AS = smem + KA * kABuf + WAVES * (kWStages * 2048) +
                                      ((QS == 1) ? stage * WAVES * SCL : ((k >> 2) & 1) * WAVES * ASB + (k & 3) * 4);
    int sb = 127;
    if constexpr (MX && QS == 1)
      sb = (int)(*reinterpret_cast<const uint32_t*>(smem + KA * kABuf + WAVES * (kWStages * 2048) + KA * WAVES * SCL +
                                                    (wave * kWStages + wstage) * SCL + nl * 4) >> (8 * kq)) & 0xff;
    if constexpr (MX && QS == 4)
      sb = (int)(*reinterpret_cast<const uint32_t*>(smem + KA * kABuf + WAVES * (kWStages * 2048) + 2 * WAVES * ASB +
                                                    (wave * 2 + ((k >> 2) & 1)) * BSB + nl * 16 + (k & 3) * 4) >> (8 * kq)) & 0xff;
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(A + mt * 2048 + pa);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(A + mt * 2048 + (pa ^ 64));
      if constexpr (INT8) {  // acc holds int32 bit patterns
        i32x4 c = __builtin_bit_cast(i32x4, acc[mt]);
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, b1), c, 0, 0, 0);
        acc[mt] = __builtin_bit_cast(f32x4, c);
      } else {
        const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        int sa = 127;
        if constexpr (MX) {  // row mt * 16 + nl sits in the region of wave row / RPW, slot row % RPW
          const int row = mt * 16 + nl;
          sa = (int)(*reinterpret_cast<const uint32_t*>(AS + (row / RPW) * ((QS == 1) ? SCL : ASB) + (row % RPW) * ((QS == 1) ? 4 : 16)) >> (8 * kq)) & 0xff;
        }
        acc[mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc[mt], 0, 0, 0, sa, 0, sb);
      }
    }
    stage = (stage == KA - 1) ? 0 : stage + 1;
    wstage = (wstage == kWStages - 1) ? 0 : wstage + 1;
  }
  };
  if constexpr (GROUPED && MT >= 4) {
    if (mt_have <= 1) k_loop(std::integral_constant<int, 1>{});
    else if (mt_have <= 2) k_loop(std::integral_constant<int, 2>{});
    else if (MT == 4 || mt_have <= 4) k_loop(std::integral_constant<int, 4>{});
    else k_loop(std::integral_constant<int, MT>{});
  } else {
    k_loop(std::integral_constant<int, MT>{});
  }
  wait_vmcnt<0>();  // the clamped fills past the end still write LDS
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  if (TRACE) ts[10] = __builtin_amdgcn_s_memtime();
  auto dump = [&] {
    if (TRACE && p.trace != nullptr && tid == 0) {
      ts[12] = __builtin_amdgcn_s_memtime();
      unsigned long long* t = p.trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
      for (int i = 0; i < 16; ++i) t[i] = ts[i];
    }
  };
  // more than four K parts (rowwise kinds): the two-level meeting of splitk.h (round 4) -- groups of four parts, so the last arriver
  // reads 3 + 3 parked tiles in two round trips instead of S - 1 in S / 4 (the 70B / TP8 qkv shard at M = 128 cuts K 16 ways: 15 x 64 KiB
  // through one CU's memory path)
  bool go_on = true;
  if (S > 1) {
    const int otile = by * gx + bx;
    int* flag = reinterpret_cast<int*>(smem);
    if constexpr (!GROUPED) {
      go_on = (S > 4) ? split_k_meet2<MT, 64 * WAVES, INT8, 4>(acc, p.ws, p.tickets, otile, S, ks, tid, flag)
                      : split_k_meet<MT, 64 * WAVES, INT8>(acc, p.ws, p.tickets, otile, S, ks, tid, flag);
    } else {
      go_on = split_k_meet<MT, 64 * WAVES, INT8>(acc, p.ws, p.tickets, otile, S, ks, tid, flag);
    }
  }
  if (!go_on) {
    dump();
    return;
  }
  if (TRACE) ts[11] = __builtin_amdgcn_s_memtime();

  // D layout: lane (col = nl, kq) holds rows 4 kq + {0..3} of each 16 x 16 tile
  if constexpr (!GROUPED) {
    // Round 5: the scaled tile goes through the (now idle) LDS and leaves as 16-byte row pieces.  The direct form below stores two bytes
    // per lane and instruction -- 32 store instructions per wave, 256 per workgroup, each a pass through the CU's one address path
    // (~5.7 k cycles of the round-4 trace) -- this one 4 per wave.  Row scales come in once per workgroup (one load per row, through LDS)
    // instead of 16 - 32 loads per lane.  Same arithmetic, same bits.
    if (p.N % 8 == 0) {
      constexpr int BNW = 16 * WAVES;            // columns of the workgroup's tile
      constexpr int RS = BNW * 2 + 16;           // staging row stride in bytes (+ 16: the 4 kq row groups of a b16 write land 8 banks apart)
      float* sa_lds = reinterpret_cast<float*>(smem + BM * RS);
      float* sb_lds = sa_lds + BM;
      float* bias_lds = sb_lds + BNW;
      __syncthreads();  // (the meeting's flag word is dead; every wave is past the loop's last LDS read)
      if (tid < BM) sa_lds[tid] = pre_sa;
      asm volatile("" : "+v"(pre_bias));  // (opaque until here: the compiler otherwise converts -- and waits for the load -- at the kernel's head)
      if (tid < BNW) { sb_lds[tid] = pre_sb; bias_lds[tid] = bf16_lo_to_f32(pre_bias); }
      __syncthreads();
      auto put = [&](int mt, int ntl, const f32x4& c) {  // m-tile mt of the slab, n-tile ntl of the workgroup's tile
        const float sbv = sb_lds[ntl * 16 + nl];
        const float bv = bias_lds[ntl * 16 + nl];
        const f32x4 sa4 = *reinterpret_cast<const f32x4*>(sa_lds + mt * 16 + kq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if constexpr (INT8) v = round_bf16((float)__builtin_bit_cast(i32x4, c)[r] * sa4[r]) * sbv;
          else v = c[r] * sa4[r] * sbv;
          if (p.bias != nullptr) v += bv;
          *reinterpret_cast<uint16_t*>(smem + (mt * 16 + kq * 4 + r) * RS + (ntl * 16 + nl) * 2) = f32_to_bf16_bits(v);
        }
      };
      if constexpr (SM) {
        const int wm = wave & 1, wn = wave >> 1;
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) put(MH * wm + i, 2 * wn + j, acc[2 * i + j]);
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) put(mt, wave, acc[mt]);
      }
      __syncthreads();
      constexpr int PPR = BNW / 8;  // 16-byte pieces per row
      uint16_t* __restrict__ yo = p.y;
#pragma unroll
      for (int it = 0; it < (BM * PPR) / (64 * WAVES); ++it) {
        const int c = it * (64 * WAVES) + tid;
        const int row = c / PPR, piece = c - row * PPR;
        const int m = m0 + row, n = bx * BNW + piece * 8;
        if (m < m_end && n + 8 <= p.N)
          *reinterpret_cast<u32x4*>(yo + (size_t)m * p.N + n) = *reinterpret_cast<const u32x4*>(smem + row * RS + piece * 16);
      }
      dump();
      return;
    }
  }
  if constexpr (SM) {
    const int wm = wave & 1, wn = wave >> 1;
    uint16_t* __restrict__ y = p.y;
    float sbj[2], biasj[2];
    int nj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int tj = bx * WAVES + 2 * wn + j;
      nj[j] = (tj < ntiles) ? tj * 16 + nl : -1;
      sbj[j] = p.scale_b[min(tj, ntiles - 1) * 16 + nl];
      biasj[j] = p.bias != nullptr ? bf16_lo_to_f32(p.bias[min(tj, ntiles - 1) * 16 + nl]) : 0.f;
    }
    float sa[16];  // all row scales first: the stores below must not sit between dependent loads
#pragma unroll
    for (int i = 0; i < 16; ++i) sa[i] = p.scale_a[min(m0 + (4 * wm + (i >> 2)) * 16 + kq * 4 + (i & 3), p.M - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + (4 * wm + i) * 16 + kq * 4 + r;
          if (m < m_end && nj[j] >= 0) {
            float v;
            if constexpr (INT8) v = round_bf16((float)__builtin_bit_cast(i32x4, acc[2 * i + j])[r] * sa[i * 4 + r]) * sbj[j];
            else v = acc[2 * i + j][r] * sa[i * 4 + r] * sbj[j];
            if (p.bias != nullptr) v += biasj[j];
            y[(size_t)m * p.N + nj[j]] = f32_to_bf16_bits(v);
          }
        }
    dump();
    return;
  }
  if (tile >= ntiles) { dump(); return; }
  const int n = tile * 16 + nl;
  uint16_t* __restrict__ y = p.y;
  if constexpr (MX) {  // scales were applied by the MFMA: out = bf16(acc)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + kq * 4 + r;
        if (m < m_end) y[(size_t)m * p.N + n] = f32_to_bf16_bits(acc[mt][r]);
      }
  } else {
    const float* __restrict__ scale_a = p.scale_a;
    const float sb = p.scale_b[(size_t)expert * p.N + n];
    const float bias = p.bias != nullptr ? bf16_lo_to_f32(p.bias[n]) : 0.f;
    float sa[4 * MT];  // all row scales first: the stores below must not sit between dependent loads
#pragma unroll
    for (int i = 0; i < 4 * MT; ++i) sa[i] = scale_a[min(m0 + (i >> 2) * 16 + kq * 4 + (i & 3), p.M - 1)];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + kq * 4 + r;
        if (m < m_end) {  // (= p.M for the ungrouped kinds)
          float v;
          if constexpr (INT8) {
            // t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))   (int8_tensor.py:315-359)
            v = round_bf16((float)__builtin_bit_cast(i32x4, acc[mt])[r] * sa[mt * 4 + r]) * sb;
          } else {
            v = acc[mt][r] * sa[mt * 4 + r] * sb;
          }
          if (p.bias != nullptr) v += bias;
          y[(size_t)m * p.N + n] = f32_to_bf16_bits(v);
        }
      }
  }
  dump();
}

// ---------------------------------------------------------------------------------------------------------------------
// Stream-K form of the MX decode kernel (round 3; 4 waves, 64-row slabs, 64-column tiles).
//
// rb8_kernel above gives every (non-empty slab, 64-column tile) its own workgroup.  At decode sizes a workgroup then lives
// for only 32 - 112 k steps, of which (s_memtime traces, profiles/mx_rb_trace_r03.txt) 3 300 - 5 600 cycles go to finding its
// group, 1 000 - 6 000 to priming the rings, 600 - 1 300 to the first data and ~2 000 to the epilogue; and the grid is whatever the
// router made it -- 672 workgroups on 512 slots (w1 with three experts hit), 192 on 256 CUs (w2).  In the k loop itself a CU
// streams 20 - 22 KiB/us (5.2 - 5.6 TB/s over the chip): the launch structure, not the loop, kept config 5 at 0.32 of HBM.
//
// Here the grid is the number of resident slots and does not depend on the routing: the (slab, tile, k step) space, G steps,
// is cut into gridDim.x equal contiguous shares.  A workgroup primes its rings ONCE and walks its share, crossing tile -- and
// expert -- boundaries with the DMA rings running: three cursors (weight issue, activation issue, compute) step through the
// same sequence at their own distance and look the next slab up in a group table held in registers (lane e: expert e's rows;
// E <= 64), so no load result is ever waited for inside the loop.  A tile that lies wholly inside a share is stored from the
// loop; a tile cut by a share boundary has its pieces parked in the split-K workspace (fp32, written through) when their
// workgroups finish -- the piece at the head of a share waits in 16 VGPRs until then -- and the last arriver adds the pieces
// in k order and stores the tile.  At most two pieces per workgroup: <= 2 x 16 KiB against ~300 KiB of weights streamed.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kStreamMinShare = 16;  // k steps: fewer would not pay for priming the rings

// QS = k steps per scale fetch.  1: one 4-byte DMA per row, step and operand (16 lanes of a wave).  4: one 16-byte DMA per row and FOUR
// steps (K % 512 == 0, 3 weight stages).  The dword DMAs moved 1 % of the bytes and cost a quarter of the kernel: with them switched
// off (timing probe, profiles/mx_rb_trace_r03.txt session F) w1 went 65.9 -> 49.3 us and w2 64.7 -> 44.9 -- the LDS-DMA path is
// bound by instructions, not bytes (fetching the activation tile through ONE line per DMA instead of eight: -4 %).
// CAST (round 6, SURVEY 8 f1 for the MX format): 0 -- p.a / p.a_mx are e4m3 codes and E8M0 scales (the caller cast the activations);
// 1 + mode -- p.a is the BF16 activation matrix and the 1 x 32 cast (to_mx, mode = AO_MX_SCALE_FLOOR / RCEIL) happens in the A-fill:
// the 16 waves fetch the step's [64 rows][128 k] bf16 tile (16 KiB: one 1 KiB LDS-DMA each, four rows per wave) ONE step ahead into a
// two-stage raw ring, and during step i every thread reads back the 16 bytes its own lane fetched for step i + 1 (no barrier: the
// wave's own vmcnt covers them), casts them with the stand-alone cast's function (quant_math.h: mx_cast8 -- four lanes per 32-block)
// and writes 8 codes into the swizzled operand tile of step i + 1 and the block's exponent byte into its scale slot; the step's barrier
// publishes both.  No cast kernel, no e4m3 copy of the activations in HBM, the same bits.
template <int WAVES, int SW, int QS, bool TRACE, int CAST = 0>
__global__ __launch_bounds__(64 * WAVES) void mx_stream_kernel(Rb8Args p) {
  static_assert(QS == 1 || (QS == 4 && SW == 3), "mx_stream_kernel: scale fetches per step, or per 4 steps with 3 weight stages");
  constexpr bool kCast = CAST != 0;
  static_assert(!kCast || (WAVES == 16 && QS == 4), "mx_stream_kernel: the fused cast is built on the 16-wave form");
  constexpr int MT = 4, BM = 64, BN = 16 * WAVES, SCL = 64, kABuf = MT * 2048, NTHR = 64 * WAVES;
  constexpr int KA = kCast ? 2 : kStages;        // stages of the e4m3 operand ring of the activations
  constexpr int kRaw = kCast ? BM * 256 : 0;     // one stage of the raw bf16 ring (two stages)
  constexpr int kWOff = KA * kABuf + 2 * kRaw;   // where the weight rings begin
  constexpr int AD = (WAVES >= 8) ? 1 : 8 / WAVES;  // activation DMAs per wave and step (8 rows each): the tile is shared by the workgroup's waves (16 waves: the first 8 fetch)
  constexpr int RPW = BM / WAVES;   // activation-scale rows fetched per wave
  constexpr int LPSC = AD + 2 + (QS == 1 ? 2 : 0);  // DMAs of one stage (activations, weights; QS == 1: + the scales of both)
  constexpr int ASB = (QS == 1) ? SCL : RPW * 16, BSB = (QS == 1) ? SCL : 256;  // bytes of one scale slot (activations per wave, weights per wave)
  constexpr int ASN = (QS == 1) ? kStages : 2, BSN = (QS == 1) ? SW : 2;          // slots per wave
  // 16 waves (round 6): ONE workgroup per CU over 256-column tiles -- half the activation re-reads per weight byte, and no second, younger
  // workgroup on the CU that the instruction arbiter serves last (the round-6 traces); its activation pieces are only requested where the
  // group has rows (QS == 4: a_cnt), so waves 8 .. 15 never issue one
  static_assert((WAVES == 8 && QS == 1) || (WAVES == 16 && QS == 4), "mx_stream_kernel: 8 waves with the scales per step, or 16 with the scales per 4 steps");
  unsigned long long ts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (TRACE) { ts[0] = __builtin_amdgcn_s_memtime(); ts[3] = __builtin_amdgcn_s_memrealtime(); }  // [3] / [4]: the 100 MHz clock at entry / exit
  // [KA][64][128 B] a | CAST: [2][64][256 B] raw bf16 | [WAVES][SW][2 KiB] b | a scales [ASN][WAVES][ASB] (CAST: [2][64 rows][4 B]) | b scales [WAVES][BSN][BSB]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  const int ksteps = p.K >> 7, NT1 = (p.N + BN - 1) / BN, NT = (p.b2 != nullptr) ? 2 * NT1 : NT1, n16 = p.N >> 4;  // NT: column tiles of one slab (both products)
  const uint32_t kb32 = (uint32_t)(p.K >> 5);

  // group table: lane e holds expert e's rows [lo, hi), its slab count and the running slab count
  int lo = 0, hi = 0;
  if (p.offs != nullptr) {
    if (lane < p.E) {  // (clamped to [0, M] and made monotone by the max below: a malformed offs cannot send a DMA out of the tensors)
      hi = min(max(p.offs[lane], 0), p.M);
      lo = (lane > 0) ? min(max(p.offs[lane - 1], 0), p.M) : 0;
    }
  } else if (lane == 0) {
    hi = p.M;
  }
  const int ns = max(hi - lo + BM - 1, 0) / BM;
  int incl = ns;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  const int excl = incl - ns;
  const int nslabs = __builtin_amdgcn_readlane(incl, 63);
  const int G = nslabs * NT * ksteps;  // < 2^31 (launcher)
  const int W = min((int)gridDim.x, max(1, G / kStreamMinShare));
  const int w = blockIdx.x;
  if (w >= W) return;  // uniform, before any DMA or barrier
  // Shares (round 6): cut at single steps (rounds 3 - 5: at multiples of QS -- at 21 steps a share that meant 24 steps for a quarter of
  // the workgroups and 20 for the rest; the scales' 4-step blocks stay aligned to the TILE, a share that begins inside one fetches it
  // whole: gv below); the first `sr` workgroups take sq + 1 steps, the others sq: ONE 32-bit division here and one per owner() (the
  // GQ * v / W of rounds 3 - 5 were 64-bit divisions, ~300 scalar instructions each, four to eight of them on every workgroup's way in
  // and out).  B(v) = where workgroup v's share begins, owner(g) = the workgroup whose share holds step g.
  // (Measured and not kept, profiles/mx_stream_ab_r06.txt: shares of equal COST -- a 64-row slab's step takes 1.10 - 1.14 x a 32-row
  // slab's -- level the workgroups' exit times and leave the launch where it was: the loop runs at the memory system's rate, and a
  // workgroup that leaves early leaves its bandwidth to the others.  The same for shares weighted by dispatch order on the two-per-CU form.)
  const unsigned sq = (unsigned)G / (unsigned)W, sr = (unsigned)G - sq * (unsigned)W;  // sq >= 16: W <= G / 16
  auto B = [&](int v) { return (int)((unsigned)v * sq + min((unsigned)v, sr)); };
  auto owner = [&](int g) {
    const unsigned big = sr * (sq + 1);
    return (int)(((unsigned)g < big) ? (unsigned)g / (sq + 1) : sr + ((unsigned)g - big) / sq);
  };
  const int g0 = B(w), g1 = B(w + 1);
  // the pieces of a tile that share boundaries cut: workgroups wf .. wf + S - 1 hold one each, parked in slot 2 v (the piece v's share
  // BEGINS with) or 2 v + 1 (the piece it ENDS with, when that is another one) -- every piece but the tile's first begins its share.
  // Of one of THIS share's cut tiles, the end that lies inside the share is this workgroup's (no division for it).
  struct Cut { int S, wf, first_odd; };
  auto cut_of = [&](int tile) {
    const int T0 = tile * ksteps;
    Cut c;
    c.wf = (T0 >= g0) ? w : owner(T0);
    c.S = ((T0 + ksteps <= g1) ? w : owner(T0 + ksteps - 1)) - c.wf + 1;
    c.first_odd = (T0 > B(c.wf)) ? 1 : 0;
    return c;
  };
  if (g0 >= g1) return;
  auto find = [&](int y, int& expert, int& m0, int& m_end) {  // y-th non-empty slab; wave-uniform, registers only
    const unsigned long long hit = __ballot(incl > y);
    const int l = __builtin_ctzll(hit);
    expert = l;
    m_end = __builtin_amdgcn_readlane(hi, l);
    m0 = __builtin_amdgcn_readlane(lo, l) + (y - __builtin_amdgcn_readlane(excl, l)) * BM;
  };
  if (TRACE) ts[13] = __builtin_amdgcn_s_memtime();

  const uint32_t a_lds = lds_offset(smem);
  const uint32_t w_lds = a_lds + kWOff + wave * (SW * 2048);
  const uint32_t as_lds = a_lds + kWOff + WAVES * (SW * 2048);
  const uint32_t bs_lds = as_lds + ASN * WAVES * ASB + wave * (BSN * BSB);
  const int tile0 = g0 / ksteps, k00 = g0 - tile0 * ksteps;

  // ---- weight cursor
  uint32_t boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * i + (lane >> 3);
    boff[i] = (uint32_t)row * (uint32_t)p.K + ((((lane & 7) ^ (row >> 1)) & 7) << 4);
  }
  const uint32_t bsoff = (uint32_t)nl * kb32;
  const uint8_t* brows = nullptr;
  const uint8_t* bsrows = nullptr;
  int gw = g0, kw = k00, tilew = tile0;
  auto set_w = [&](int tile) {
    const int slab = tile / NT, nt = tile - slab * NT;
    int e, m0, m_end;
    find(slab, e, m0, m_end);
    const bool second = nt >= NT1;  // (the second product's tiles)
    const int t16 = min((second ? nt - NT1 : nt) * WAVES + wave, n16 - 1);  // tiles past N alias the last one; never stored
    brows = (second ? p.b2 : p.b) + ((size_t)e * p.N + (size_t)t16 * 16) * p.K;
    bsrows = (second ? p.b2_mx : p.b_mx) + ((size_t)e * p.N + (size_t)t16 * 16) * kb32;
  };
  auto issue_w = [&](int stage) {  // the cursor's step into `stage`, then on to the next step (the last step repeats past the end)
    dma_b128_nt_s(brows + (size_t)kw * 128, boff[0], w_lds + stage * 2048);
    dma_b128_nt_s(brows + (size_t)kw * 128, boff[1], w_lds + stage * 2048 + 1024);
    if constexpr (QS == 1) {
      if (lane < 16) dma_b32_s(bsrows + (size_t)kw * 4, bsoff, bs_lds + stage * SCL);
    }
    if (gw < g1 - 1) {
      ++gw;
      if (++kw == ksteps) { kw = 0; set_w(++tilew); }
    }
  };
  // ---- activation cursor
  // QS == 4: a wave only requests the 8-row pieces (and the scale rows) its group HAS -- a 32-row group costs 4 of the tile's 8
  // DMA instructions, and on this path instructions are what costs.  The hand-counted waits then follow what was really
  // issued (a_prev / s_prev below); rows nobody fetched keep stale bytes that only the m-tiles past the group would read.
  uint32_t aoff[AD], asoff = 0;
  int a_cnt = AD;  // pieces of the cursor's slab this wave fetches
  int ga = g0, ka = k00, tilea = tile0;
  auto set_a = [&](int tile) {
    int e, m0, m_end;
    find(tile / NT, e, m0, m_end);
#pragma unroll
    for (int i = 0; i < AD; ++i) {
      const int row = 8 * (AD * wave + i) + (lane >> 3);  // rows past the group re-read its last row (one line, never used)
      aoff[i] = (uint32_t)min(m0 + row, m_end - 1) * (uint32_t)p.K + ((((lane & 7) ^ (row >> 1)) & 7) << 4);
    }
    asoff = (uint32_t)min(m0 + RPW * wave + (lane % RPW), m_end - 1) * kb32;
    if constexpr (QS == 4) a_cnt = __builtin_amdgcn_readfirstlane(max(0, min(AD, (min(m_end - m0, BM) - 8 * AD * wave + 7) >> 3)));
    if constexpr (kCast) {  // the raw bf16 tile: wave w fetches rows 4 w .. 4 w + 3 (lane: row 4 w + lane / 16, 16-byte chunk lane % 16 of its 256 B)
      const int row = 4 * wave + (lane >> 4);
      aoff[0] = (uint32_t)min(m0 + row, m_end - 1) * (uint32_t)p.K * 2u + ((lane & 15) << 4);
      a_cnt = __builtin_amdgcn_readfirstlane((4 * wave < min(m_end - m0, BM)) ? 1 : 0);
    }
  };
  auto issue_a = [&](int stage) -> int {  // returns the DMAs issued
    const int cnt = a_cnt;
    if constexpr (kCast) {  // (stage: the RAW ring's)
      if (cnt) dma_b128_s(p.a + (size_t)ka * 256, aoff[0], a_lds + KA * kABuf + stage * kRaw + wave * 1024);
    } else {
#pragma unroll
    for (int i = 0; i < AD; ++i)
      if (QS == 1 || i < cnt) dma_b128_s(p.a + (size_t)ka * 128, aoff[i], a_lds + stage * kABuf + (AD * wave + i) * 1024);
    }
    if constexpr (QS == 1) {
      if (lane < RPW) dma_b32_s(p.a_mx + (size_t)ka * 4, asoff, as_lds + (stage * WAVES + wave) * SCL);
    }
    if (ga < g1 - 1) {
      ++ga;
      if (++ka == ksteps) { ka = 0; set_a(++tilea); }
    }
    return cnt;
  };
  // ---- scale cursor (QS == 4): the 16 scale bytes of 4 steps per row, two slots per wave and operand
  uint32_t asoff4 = 0;
  bool as_have = true;  // this wave's activation-scale rows exist in the cursor's slab
  const uint8_t* bsrows4 = nullptr;
  const int gv = g0 - (k00 & 3);  // QS == 4: the step the share's first 4-step scale block begins with (tiles begin at multiples of 4)
  int gs = gv, ks4 = k00 & ~3, tiles4 = tile0;
  auto set_s = [&](int tile) {
    const int slab = tile / NT, nt = tile - slab * NT;
    int e, m0, m_end;
    find(slab, e, m0, m_end);
    const bool second = nt >= NT1;
    const int t16 = min((second ? nt - NT1 : nt) * WAVES + wave, n16 - 1);
    bsrows4 = (second ? p.b2_mx : p.b_mx) + ((size_t)e * p.N + (size_t)t16 * 16) * kb32;
    asoff4 = (uint32_t)min(m0 + RPW * wave + (lane % RPW), m_end - 1) * kb32;
    as_have = !kCast && RPW * wave < m_end - m0;  // (CAST: the activation scales are computed here, not fetched)
  };
  auto issue_s = [&](int slot) -> int {  // (the last block repeats past the end, like the tiles); returns the DMAs issued
    const int cnt = as_have ? 2 : 1;
    if (as_have) {
      if (lane < RPW) dma_b128_s(p.a_mx + (size_t)ks4 * 4, asoff4, as_lds + (slot * WAVES + wave) * ASB);
    }
    if (lane < 16) dma_b128_s(bsrows4 + (size_t)ks4 * 4, bsoff, bs_lds + slot * BSB);
    if (gs + 4 < g1) {
      gs += 4;
      if ((ks4 += 4) == ksteps) { ks4 = 0; set_s(++tiles4); }
    }
    return cnt;
  };
  auto wait_upto = [&](int n) {  // s_waitcnt vmcnt(n) for a wave-uniform n in [2, AD + 5] (the immediate has to be a constant)
    switch (n) {
      case 2: wait_vmcnt<2>(); break;
      case 3: wait_vmcnt<3>(); break;
      case 4: wait_vmcnt<4>(); break;
      case 5: wait_vmcnt<5>(); break;
      case 6: wait_vmcnt<6>(); break;
      default: wait_vmcnt<7>(); break;
    }
  };
  // ---- compute cursor
  int kc = k00, kb = k00, tilec = tile0, m0c = 0, m_endc = 0, ntc = 0, mt_have = 0;
  auto set_c = [&](int tile) {
    const int slab = tile / NT;
    int e;
    find(slab, e, m0c, m_endc);
    ntc = tile - slab * NT;
    mt_have = __builtin_amdgcn_readfirstlane((min(m_endc - m0c, BM) + 15) >> 4);
  };
  set_w(tile0);
  set_a(tile0);
  set_c(tile0);
  if constexpr (QS == 4) set_s(tile0);

  auto store_tile = [&](const f32x4 (&v)[MT], int m0, int m_end, int nt) {  // scales were applied by the MFMA: out = bf16(acc)
    const bool second = nt >= NT1;
    const int t16 = (second ? nt - NT1 : nt) * WAVES + wave;
    if (t16 >= n16) return;
    uint16_t* __restrict__ y = second ? p.y2 : p.y;
    const int n = t16 * 16 + nl;  // D layout: lane (col = nl, kq) holds rows 4 kq + {0..3} of each 16 x 16 tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + kq * 4 + r;
        if (m < m_end) y[(size_t)m * p.N + n] = f32_to_bf16_bits(v[mt][r]);
      }
  };

  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // pieces of cut tiles are parked [slot][m-tile][thread] (slot 2 v: the piece workgroup v's share begins with, 2 v + 1: the piece
  // it ends with when that is another one); fp32, written through (sc1): the readers sit on other XCDs
  constexpr int kSc1 = 16;
  constexpr int kRegBytes = NTHR * 16, kPartBytes = MT * kRegBytes;
  const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, 0x7fffffff, 0x00020000);
  auto park = [&](const f32x4 (&v)[MT], int tile, int mth) {  // mth: m-tiles the tile's group has (only those are parked and read)
    const int mine = (2 * w + ((tile * ksteps > g0) ? 1 : 0)) * kPartBytes;  // same rule as the reader's (cut_of)
#pragma unroll
    for (int r = 0; r < MT; ++r)
      if (r < mth) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[r]), rws, tid * 16 + r * kRegBytes, mine, kSc1);
    // the data registers stay live (and untouched) for 16 more cycles: the caller zeroes them next, and a buffer_store_dwordx4 with an
    // SGPR soffset whose data register is overwritten by the next instruction can store the new value on gfx950 (splitk.h)
    asm volatile("s_nop 7\n\ts_nop 7" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  };
  bool has_head = false;
  int head_tile = 0, head_m0 = 0, head_mend = 0, head_nt = 0, head_mth = 0;
  // round 6 (see the meeting below): the head piece's ticket is taken from the loop `head_wait` steps after it was parked (wave 1, lane 0:
  // head_t; `lenient` = the steps whose wait lets that one returning atomic stay in flight); the tail tile's ticket word is read two steps
  // before the share ends.  p.ablate (A/B): 1 head ticket after the loop, 2 no early read -- the round-3 protocol.
  int head_wait = 0, lenient = 0;
  unsigned head_t = 0u, peek = 0xffffffffu;
  const int tile_last = (g1 - 1) / ksteps;
  const bool tail_is_piece = !(tile_last * ksteps >= g0 && (tile_last + 1) * ksteps == g1);
  const int g_peek = (tail_is_piece && g1 - g0 >= 3 && !(p.ablate & 2)) ? g1 - 2 : -1;
  const __amdgpu_buffer_rsrc_t rtk = __builtin_amdgcn_make_buffer_rsrc(p.tickets, 0, 0x7fffffff, 0x00020000);
  const int pa = nl * 128 + (((kq ^ (nl >> 1)) & 7) << 4);

  // Issue order (SW >= 4): w(0 .. SW-4) | a(0) w(SW-3) | a(1) w(SW-2), then per step a(i+2) w(i+SW-1): when step i starts the youngest
  // requests are a(i+1), w(i+SW-2) (one stage: LPSC) and w(i+SW-3) (3 DMAs); everything older has landed.  SW == 3: a(0) w(0) |
  // a(1) w(1), per step a(i+2) w(i+2): one stage may be in flight.  (The stores of a tile finished inside the loop are younger
  // than what the next two waits need and VMEM retires in order: those waits only become stricter, never wrong.)
  if (TRACE) ts[14] = __builtin_amdgcn_s_memtime();
  // QS == 4: the scales of the block the share begins in first; of each next block at the second step of the one before (below) -- a share
  // that begins at the third or fourth step of its block is past that point: its second block goes out here too
  if constexpr (QS == 4) {
    issue_s(0);
    if ((k00 & 3) >= 2) issue_s(1);
  }
#pragma unroll
  for (int i = 0; i < SW - 3; ++i) issue_w(i);
  [[maybe_unused]] const int a_first = issue_a(0);
  issue_w(SW - 3);
  int a_prev = issue_a(1), s_prev = 0;  // QS == 4: what the step before issued besides its two weight DMAs (= all that may be in flight)
  issue_w(SW - 2);
  // CAST: this thread's 16 bytes of raw stage `rs` (fetched by its own lane) -> 8 codes in operand stage `os` + the block's exponent byte
  [[maybe_unused]] auto convert = [&](int rs, int os) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + KA * kABuf + rs * kRaw + wave * 1024 + lane * 16);
    uint32_t e;
    const u32x2 q = mx_cast8<(CAST > 0 ? CAST - 1 : 0)>(v, e);
    const int row = 4 * wave + (lane >> 4), c = lane & 15, r = row & 15;  // c: which 8 of the step's 128 k
    *reinterpret_cast<u32x2*>(smem + os * kABuf + (row >> 4) * 2048 + r * 128 + ((((c >> 1) ^ (r >> 1)) & 7) << 4) + (c & 1) * 8) = q;
    if ((lane & 3) == 0) *reinterpret_cast<uint8_t*>(smem + kWOff + WAVES * (SW * 2048) + os * 256 + row * 4 + (c >> 2)) = (uint8_t)e;
  };
  if constexpr (kCast) {
    // raw(0) has to be cast before the first barrier: everything up to it has landed when at most raw(1) and the two weight stages behind
    // it are in flight
    wait_upto(4 + a_prev);
    if (a_first) convert(0, 0);
  }
  if (TRACE) ts[1] = __builtin_amdgcn_s_memtime();
  // (the share's cut tiles are worked out where they are needed: the tail's at the step that reads its ticket word, the head's by the one
  // lane that holds its ticket -- ahead of the loop they cost every workgroup 2 - 4 k cycles between priming and the first step)
  Cut cut_tail{1, w, 0};
  int stage = 0, wstage = 0;
  for (int g = g0; g < g1; ++g) {
    // (CAST: raw(i + 1), requested first in the step before, has to be in: only that step's two weight DMAs may still be in flight)
    if constexpr (kCast) { wait_upto(2 + (lenient > 0 ? 1 : 0)); lenient = max(lenient - 1, 0); }
    else if constexpr (QS == 4) { wait_upto(a_prev + 2 + s_prev + (lenient > 0 ? 1 : 0)); lenient = max(lenient - 1, 0); }
    else if constexpr (SW >= 4) wait_vmcnt<LPSC + 3>(); else wait_vmcnt<LPSC>();
    // everyone's share of the activation tile has landed, and everyone has finished reading the step before
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (TRACE && g == g0) ts[2] = __builtin_amdgcn_s_memtime();  // static indices: the stamps stay in SGPRs
    if (TRACE && g == g0 + 7) ts[9] = __builtin_amdgcn_s_memtime();
    if constexpr (QS == 4) {
      // step i counted from the share's first scale block (i = g - gv; tiles begin at multiples of 4): the next block's scales go out at i % 4 == 1 into
      // the slot the block before this one used (last read at step i - 2); they are older than a(i + 2), w(i + 2), whose wait at
      // step i + 2 therefore covers them, and at that wait -- only there -- two more requests are younger than what it needs
      s_prev = (((g - gv) & 3) == 1) ? issue_s((((g - gv) >> 2) + 1) & 1) : 0;
    }
    [[maybe_unused]] const int a_landed = a_prev;  // CAST: whether this wave fetched rows of step i + 1's tile (requested a step ago)
    a_prev = issue_a(kCast ? stage : (stage == 0) ? 2 : stage - 1);  // (CAST: raw(i + 2) into the raw stage cast a step ago)
    issue_w((wstage == 0) ? SW - 1 : wstage - 1);
    if constexpr (kCast) {
      if (a_landed) convert(stage ^ 1, stage ^ 1);
    }
    // (both behind this step's DMAs: the next wait then only asks the oldest of them to have landed one step early)
    if (head_wait > 0 && --head_wait == 0) {  // uniform.  The park's stores of every wave have retired (two waits, two barriers since)
      if (wave == 1) {
        if (lane == 0) head_t = __hip_atomic_fetch_add(&p.tickets[head_tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The atomic may stay in flight across the waits in which it is YOUNGER than what the wait needs, or sits between what it needs and
        // what it allows (QS == 1: the static waits ask for it, at worst a stall).  Without the cast that is two waits: step i + 1 needs the
        // stage requested in step i - 1 (everything of step i + the atomic may fly), step i + 2 needs step i's (older than the atomic: with
        // step i + 1's requests AND the atomic allowed, everything older than the atomic has still landed).  With the cast only ONE: step
        // i + 2 needs raw(i + 3), requested in step i + 1 -- YOUNGER than the atomic; with three in flight allowed its two weight DMAs and
        // raw(i + 3) itself could be the three.  Round 6 shipped 2 there: wave 1 then cast rows 4 .. 7 of a tile from a raw stage that had
        // not landed in ~1 % of the launches whose shares begin inside a tile (tools/fuzz_long.py, tools/stress_mx_pair.py;
        // profiles/mx_pair_race_r06.jsonl), one k step of garbage in 4 rows x 256 columns.
        if constexpr (QS == 4) lenient = kCast ? 1 : 2;
      }
    }
    if (g == g_peek) {
      if (wave == 0) peek = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rtk, 0, tile_last * 4, kSc1);
      cut_tail = cut_of(tile_last);
    }
    const char* A = smem + stage * kABuf;
    const char* Wt = smem + kWOff + (wave * SW + wstage) * 2048;
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(Wt + pa);  // the n-tile's 16 rows are laid out like an m-tile
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(Wt + (pa ^ 64));
    const i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    // the scale byte of lane group kq is that of 32-k block kq of the step (operand layout probed on gfx950, stream8_kernels.hip)
    const int blk = ((g - gv) >> 2) & 1, ph = (g - gv) & 3;  // QS == 4: slot and position of this step's scale dword
    const char* AS = smem + kWOff + WAVES * (SW * 2048) + (kCast ? stage * 256 : ((QS == 1) ? stage : blk) * WAVES * ASB + ((QS == 1) ? 0 : ph * 4));
    const char* BS = smem + kWOff + WAVES * (SW * 2048) + ASN * WAVES * ASB + (wave * BSN + ((QS == 1) ? wstage : blk)) * BSB;
    const int sb = (int)(*reinterpret_cast<const uint32_t*>(BS + ((QS == 1) ? nl * 4 : nl * 16 + ph * 4)) >> (8 * kq)) & 0xff;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt < mt_have) {  // uniform: a 32-row group reads and multiplies two m-tiles
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(A + mt * 2048 + pa);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(A + mt * 2048 + (pa ^ 64));
        const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        const int row = mt * 16 + nl;  // its scales sit in the slot of wave row / RPW (CAST: [row][4 B])
        const int sa = (int)(*reinterpret_cast<const uint32_t*>(AS + (kCast ? row * 4 : (row / RPW) * ASB + (row % RPW) * ((QS == 1) ? 4 : 16))) >> (8 * kq)) & 0xff;
        acc[mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc[mt], 0, 0, 0, sa, 0, sb);
      }
    }
    stage = (stage == KA - 1) ? 0 : stage + 1;
    wstage = (wstage == SW - 1) ? 0 : wstage + 1;
    ++kc;
    if (kc == ksteps && g + 1 < g1) {  // a tile ends inside the share
      if (kb == 0) {
        store_tile(acc, m0c, m_endc, ntc);
      } else {  // the share began inside this tile: its piece is parked now, its ticket is taken after the loop
        park(acc, tilec, mt_have);
        has_head = true; head_tile = tilec; head_m0 = m0c; head_mend = m_endc; head_nt = ntc; head_mth = mt_have;
        head_wait = (p.ablate & 1) ? (1 << 30) : 2;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      kc = 0; kb = 0;
      set_c(++tilec);
    }
  }
  wait_vmcnt<0>();  // the repeated fills past the end still write LDS; the in-loop ticket and the peek (below) have returned
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (TRACE) ts[10] = __builtin_amdgcn_s_memtime();

  // The pieces of a cut tile meet: every piece parked and written through, a ticket per piece, the last arriver of a tile adds its pieces
  // in k order and stores it.  Round 6 (the round-6 traces: the tail -- park, drain, ticket, gather, store: three dependent round trips under
  // load -- was 7.5 k ticks at the median and 15 - 21 k for the last workgroups to leave, 3 - 10 us of a 51 us launch):
  //  * the HEAD piece (the end of a tile the share began in) is parked from the loop, and its ticket is taken from the loop too, two steps
  //    later: the hand-counted waits of those two steps have retired the park's stores of every wave (VMEM retires in order) and the step's
  //    barrier has seen them all -- nothing of it is left for the tail unless it turns out to be the tile's last arriver (it is the piece
  //    that is done a whole share before the others);
  //  * the TAIL piece: two steps before the end the ticket word is read (sc1).  If every other piece had arrived by then, this workgroup is
  //    the last arriver whatever happens next: it does not park, takes no ticket, adds the others' pieces and its registers in k order and
  //    stores -- one round trip instead of three.  Otherwise the round-3 protocol: park, drain, ticket, and the last arriver gathers.
  // v = the tile's pieces added in k order; the piece of workgroup `self` (-1: none) comes from `mine` (registers) instead of the workspace
  auto gather = [&](f32x4 (&v)[MT], const Cut& c, int mth, int self, const f32x4 (&mine)[MT]) {
    f32x4 sum[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r) sum[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < c.S; q0 += 4) {  // four pieces in flight; indices past S re-read the last piece and are not added
      f32x4 x[4][MT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = min(q0 + u, c.S - 1), vq = c.wf + q;
        const int slot = (2 * vq + (q == 0 ? c.first_odd : 0)) * kPartBytes;
#pragma unroll
        for (int r = 0; r < MT; ++r) {
          x[u][r] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (vq == self) x[u][r] = mine[r];  // uniform
          else if (r < mth) x[u][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, tid * 16 + r * kRegBytes, slot, kSc1));
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool keep = q0 + u < c.S;
#pragma unroll
        for (int r = 0; r < MT; ++r) {
          sum[r].x += keep ? x[u][r].x : 0.f; sum[r].y += keep ? x[u][r].y : 0.f;
          sum[r].z += keep ? x[u][r].z : 0.f; sum[r].w += keep ? x[u][r].w : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < MT; ++r) v[r] = sum[r];
  };
  const bool tail_whole = (kb == 0 && kc == ksteps);  // the share's last tile: whole (never cut) or a piece
  bool tail_meets = false;
  if (!tail_whole && g_peek < 0) cut_tail = cut_of(tile_last);  // (a share of fewer than three steps, or the round-3 protocol: not worked out in the loop)
  if (tail_whole) {
    store_tile(acc, m0c, m_endc, ntc);
  } else {
    // ONE wave's reading decides for the workgroup (the waves issued their loads at different instants of the step: one may have seen
    // S - 2 arrivals and its neighbour S - 1 -- the round-6 build that let every wave decide for itself stored a few torn tiles per launch)
    unsigned* seen_lds = reinterpret_cast<unsigned*>(smem);  // (every wave is past the loop's last LDS read: the barrier above)
    if (tid == 0) seen_lds[2] = peek;
    __syncthreads();
    const unsigned seen = seen_lds[2];  // 0xffffffff: not read (a share of fewer than three steps)
    if (seen == (unsigned)(cut_tail.S - 1)) {  // everyone else's piece is parked, written through and ticketed: this one is the last arriver
      gather(acc, cut_tail, mt_have, w, acc);
      store_tile(acc, m0c, m_endc, ntc);
      if (tid == 0) __hip_atomic_store(&p.tickets[tilec], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    } else {
      park(acc, tilec, mt_have);
      tail_meets = true;
    }
  }
  const bool head_late = has_head && head_wait > 0;  // parked in the share's last two steps: its ticket is still to be taken
  if (has_head || tail_meets) {
    if (tail_meets || head_late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // written through before the tickets are taken
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (lane == 0 && wave < 2) {
      int last = 0;
      if ((wave == 0) ? tail_meets : has_head) {
        const int tile = (wave == 0) ? tilec : head_tile;
        unsigned t = head_t;  // (wave 1, lane 0: the ticket taken from the loop)
        if (wave == 0 || head_late) t = __hip_atomic_fetch_add(&p.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned S = (unsigned)((wave == 0) ? cut_tail.S : cut_of(head_tile).S);
        last = (t == S - 1);
        if (last) __hip_atomic_store(&p.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      }
      flag[wave] = last;
    }
    __syncthreads();
    if (flag[0]) { gather(acc, cut_tail, mt_have, -1, acc); store_tile(acc, m0c, m_endc, ntc); }
    if (flag[1]) { gather(acc, cut_of(head_tile), head_mth, -1, acc); store_tile(acc, head_m0, head_mend, head_nt); }
  }
  if (TRACE && p.trace != nullptr && tid == 0) {
    ts[11] = ts[12] = __builtin_amdgcn_s_memtime();
    ts[4] = __builtin_amdgcn_s_memrealtime();
    ts[15] = (unsigned long long)(g1 - g0);  // steps of this share
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 16;
    for (int i = 0; i < 16; ++i) t[i] = ts[i];
  }
}

thread_local unsigned long long* g_fp8_rb_trace = nullptr;  // profiling only (ao_int4_set_trace shares the pointer)
// A/B knobs of the rowwise weight-streaming kernel (ao_gemm8_set_tuning; 0 = product rule): column-tile width, K parts, same-XCD meeting
thread_local int g_rb8_bn = 0, g_rb8_split = 0, g_rb8_ablate = 0;
thread_local int g_rb8_bm = 0;  // A/B (ao_gemm8_set_tuning key 3): 64 / 128 = that slab height at any M (0: the cost model's pick)
thread_local bool g_rb8_sm = true;  // rb8_kernel's 2 x 4 wave arrangement where it is built (ao_gemm8_set_variant 103: off)

template <int WAVES, int KIND, int MT = 8, bool SLIM = false, int QS = 1>
int launch_rb8(Rb8Args p, int split, hipStream_t stream) {
  constexpr int BN = WAVES * 16;
  constexpr int BM = 16 * MT;
  constexpr bool kGrouped = (KIND == RB8_MX || KIND == RB8_FP8_GROUPED);
  // grouped: at most ceil(M / BM) + E non-empty slabs exist whatever the group sizes are (and no more than E x slabs-per-group)
  const unsigned gy = !kGrouped ? (unsigned)((p.M + BM - 1) / BM)
                      : (p.offs == nullptr) ? (unsigned)p.slabs
                                            : (unsigned)std::min<int64_t>((int64_t)p.slabs * p.E, (p.M + BM - 1) / BM + p.E);
  dim3 grid((unsigned)((p.N + BN - 1) / BN), gy, (unsigned)split), block(64 * WAVES);
  constexpr int kWStages = w_stages(WAVES, KIND, MT, SLIM);
  constexpr int KA = a_stages(WAVES, KIND, MT);
  constexpr size_t smem = (size_t)KA * MT * 2048 + (size_t)WAVES * kWStages * 2048 +
                          ((KIND != RB8_MX) ? 0 : (QS == 4) ? (size_t)2 * 16 * MT * 16 + (size_t)WAVES * 2 * 256
                                                            : (size_t)(KA + kWStages) * WAVES * (SLIM ? 64 : 256));
  static_assert(!SLIM || 3 * smem <= 160 * 1024, "SLIM: three workgroups per CU");
  static_assert(smem <= 160 * 1024, "rb8_kernel: LDS");
  if (split > 1) {
    // (two-level meeting for > 4 parts of the rowwise kinds: S + ceil(S / 4) parked tiles and 1 + ceil(S / 4) tickets per output tile)
    const bool two_level = !kGrouped && split > 4;
    const int64_t slots = two_level ? split + (split + 3) / 4 : split, tks = two_level ? 1 + (split + 3) / 4 : 1;
    AO_REQUIRE((int64_t)grid.x * grid.y * slots * BN * BM <= (int64_t)kSplitMaxTiles * 128 * 128, "rb8: %u x %u tiles x %d parts exceed the split-K workspace",
               grid.x, grid.y, split);
    AO_REQUIRE((int64_t)grid.x * grid.y * tks <= kSplitMaxTickets - 8, "rb8: %u x %u output tiles exceed the split-K tickets", grid.x, grid.y);
    if (int rc = splitk_workspace(stream, &p.ws, &p.tickets, (size_t)grid.x * grid.y * slots * BN * BM, split)) return rc;
  }
  p.trace = g_fp8_rb_trace;
  p.ablate = g_rb8_ablate;
  auto kern = (p.trace != nullptr) ? rb8_kernel<WAVES, KIND, MT, true, SLIM, QS> : rb8_kernel<WAVES, KIND, MT, false, SLIM, QS>;
  if constexpr (WAVES == 8 && MT == 8 && (KIND == RB8_FP8 || KIND == RB8_INT8) && !SLIM && QS == 1) {
    // the 2 x 4 wave arrangement (fewer operand fragments per MFMA); ao_gemm8_set_variant(103): the 1 x 8 form, for A/B
    if (g_rb8_sm) kern = (p.trace != nullptr) ? rb8_kernel<WAVES, KIND, MT, true, SLIM, QS, true> : rb8_kernel<WAVES, KIND, MT, false, SLIM, QS, true>;
  }
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(rb8_kernel)")) return rc;
  ao::launch(kern, grid, block, smem, stream, p);
  AO_LAUNCH_CHECK("rb8_kernel launch");
  return AO_OK;
}

thread_local int g_mx_proto = 0;  // A/B (ao_gemm8_set_tuning key 9): 1 head ticket after the loop, 2 no early read of the tail ticket (1 | 2: the round-3 meeting)
// Launch of the stream-K form: one workgroup per resident slot of the chip.
constexpr int kChipCUs = 256;  // MI355X
template <int WAVES, int SW, int QS, int CAST = 0>
int launch_mx_stream(Rb8Args p, hipStream_t stream) {
  constexpr size_t scales = (QS == 1) ? (size_t)(kStages + SW) * WAVES * 64 : (size_t)2 * WAVES * (64 / WAVES) * 16 + (size_t)WAVES * 2 * 256;
  constexpr size_t smem = (CAST ? (size_t)2 * 4 * 2048 + (size_t)2 * 64 * 256 : (size_t)kStages * 4 * 2048) + (size_t)WAVES * SW * 2048 + scales;
  static_assert(smem <= 160 * 1024, "mx_stream_kernel: LDS");
  constexpr int per_cu = (int)((160 * 1024) / smem);
  static_assert(per_cu >= 2 || (WAVES == 16 && per_cu == 1), "mx_stream_kernel: two workgroups per CU (one of 16 waves)");
  const unsigned Wg = (unsigned)(per_cu * kChipCUs);
  if (int rc = splitk_workspace(stream, &p.ws, &p.tickets, (size_t)2 * Wg * 64 * 16 * WAVES)) return rc;
  p.trace = g_fp8_rb_trace;
  p.ablate = g_mx_proto;  // the meeting protocol's A/B bits
  auto kern = (p.trace != nullptr) ? mx_stream_kernel<WAVES, SW, QS, true, CAST> : mx_stream_kernel<WAVES, SW, QS, false, CAST>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(mx_stream_kernel)")) return rc;
  ao::launch(kern, dim3(Wg), dim3(64 * WAVES), smem, stream, p);
  AO_LAUNCH_CHECK("mx_stream_kernel launch");
  return AO_OK;
}

thread_local bool g_mx_quad = true;  // (ao_gemm8_set_variant 129: off) block scales fetched per 4 k steps where K % 512 == 0
thread_local int g_mx_stream = 1;  // (ao_gemm8_set_variant) 1 product: the stream-K kernel for decode-size groups; 0 never (113: one workgroup per tile, the form of larger groups)
thread_local int g_fp8_rb_force = 0;  // profiling only: 0 product heuristic, 1 never, 2 always, 3 always + 64-column tiles, two workgroups per CU

}  // namespace

void fp8_rowwise_rb_set_mode(int mode) { g_fp8_rb_force = mode; }
void rb8_set_wave_grid(bool two_by_four) { g_rb8_sm = two_by_four; }
void rb8_set_tuning(int bn, int split, int ablate) { g_rb8_bn = bn; g_rb8_split = split; g_rb8_ablate = ablate; }
void rb8_set_slab_rows(int rows) { g_rb8_bm = rows; }
void mx_stream_set_tuning(int proto) { g_mx_proto = proto; }
void mx_rb_set_stream(int mode, bool quad) { g_mx_stream = mode; g_mx_quad = quad; }
void fp8_rowwise_rb_set_trace(unsigned long long* p) { g_fp8_rb_trace = p; }
bool fp8_rowwise_rb_forced() { return g_fp8_rb_force >= 2; }

// (Round 5 also built 256-row slabs: ahead of the tile kernels where 128-row slabs needed a second round of the chip, behind the 256 x 128
// phase-interleaved GEMM built later that round, which takes exactly those shapes -- removed in round 6; profiles/midm_forms_r05.jsonl.)
bool gemm8_p8h_band(int64_t M, int64_t N, int64_t K);  // gemm8_p8_kernels.hip

// True when this kernel is the better choice: the 128 x 128 GEMM grid would leave most of the chip idle (same rule for int8).
bool fp8_rowwise_rb_preferred(int64_t M, int64_t N, int64_t K) {
  if (K % 128 != 0 || N % 16 != 0 || M * K >= (1ll << 32) || N * K >= (1ll << 32)) return false;
  if (g_fp8_rb_force == 1) return false;
  if (g_fp8_rb_force >= 2) return true;
  // up to one 128 x 128 workgroup per CU (one round of the chip).  Round 4 (profiles/fp8_dispatch_sweep_r04.txt, cold 70B / TP8 shards): the bound
  // was 190, and the 224 - 256 tiles of M = 512 fell to the 4-wave two-stage tile kernel -- gate_up 73.6 us against 44.3 here (hipBLASLt
  // 47.5), down 39.8 / 23.6 (24.4), o 20.9 / 11.7 (11.5); from two rounds on (M = 1024: 448 - 512 tiles) the tiled kernels are level or ahead
  // (round 5: minus the shapes the 256 x 128 phase-interleaved GEMM takes)
  return ((N + 127) / 128) * ((M + 127) / 128) <= 256 && !gemm8_p8h_band(M, N, K);
}

// Round 6, shapes no rule had been fitted on (Llama-2-13B, Qwen2-7B, Llama-3-70B / TP4; profiles/other_shapes_forms_r06.jsonl, cold weights, fp8):
// what the decode kernels (dec8 / mid8) do not take at 8 .. 64 rows used to fall to the per-tile streaming kernel (up to 32 rows, and up to 64
// when the weight has more than 256 column tiles).  On weights of 16 MB and more this kernel is ahead there -- gate_up 37888 x 3584 at
// M = 24 / 32 / 48 / 64: 52 / 58 / 74 / 90 us -> 37 / 38 / 38 / 40; down 5120 x 13824 at M = 8 .. 16: 31 - 34 -> 22 - 23; down 3584 x 18944 at
// M = 24 / 32 (once mid8 refuses a K it cannot split): 34 -> 23; qkv 4608 x 3584 at 24 / 32: 12.7 / 13.8 -> 10.3; the 70B / TP8 down shard
// 8192 x 3584 at 24 / 32: 13.7 / 14.2 -> 13.0 / 12.6 -- and level or behind on smaller ones (o 3584 x 3584: 8.8 against 9.6), which stay.
bool rb8_small_m_preferred(int64_t M, int64_t N, int64_t K) {
  if (g_fp8_rb_force == 1) return false;
  if (K % 128 != 0 || N % 16 != 0 || M * K >= (1ll << 32) || N * K >= (1ll << 32)) return false;
  return M >= 8 && M <= 64 && N * K >= 16000000;
}

namespace {

// The slab height, tile width and K split of a rowwise launch.  Rounds 1-4 took 128-column tiles where they gave ~half a chip of workgroups
// and 64-column ones otherwise, with K cut until the grid approached 256 workgroups (up to 16 parts).  Round 5 swept (width, parts) next
// to hipBLASLt and priced the choice as a trade between three costs, in microseconds:
//   * the k loop: steps per workgroup x a per-step cost by tile width -- one wave's barrier + issue + fragment-read + MFMA chain per
//     step -- but never less than the weight bytes at the rate the stream reaches;
//   * the meeting of the K parts: one level for 2 - 4 parts, two levels from 5 on, + the last arriver's gather per extra part;
//   * a fixed cost per round of the chip (launch, priming, first data, epilogue), and rounds = ceil(workgroups / 256).
// Narrow tiles re-stage the activation tile once per tile (more steps in total); wide tiles need more K parts to fill the chip.
// Round 6 (verdict item 5): 64-ROW slabs above 64 rows as well -- M is cut instead of K: two slabs re-read a weight tile through L2, the
// parked tiles halve, half as many K parts fill the chip (often none: no meeting at all), a step stages and multiplies half the rows.
// Measured on the (slab, width, parts) grid of profiles/rb8_grid_r06_{fp8,int8}.jsonl (8 shapes x M = 80 .. 512, cold weights, 36 forms
// per cell): 64-row slabs win 59 of 84 cells, by 5 - 25 % on the TP = 8 shards at M = 128 (qkv 14.6 -> 12.3 us, o 9.0 -> 7.1, down
// 17.0 -> 15.0; gate_up keeps 128 rows).  The constants below are a least-squares fit of this model to those 958 measurements, refined
// within +- 30 % for the least regret of the pick: 0.6 % over the per-cell best form on average (the 128-row-only plan of round 5: 9.4 %).
struct Rb8Plan { int bm, bn, split; };
struct Rb8Cost { double fixed, step[3], meet1, meet2, gather; };  // step: 32 / 64 / 128 columns
constexpr Rb8Cost kRb8Cost128{4.335, {0.407, 0.465, 0.589}, 4.118, 7.041, 0.701};
constexpr Rb8Cost kRb8Cost64{2.348, {0.319, 0.330, 0.388}, 3.606, 5.628, 1.362};
constexpr double kRb8StreamBytesPerUs = 4.603e6;  // weight bytes per microsecond the loop sustains (4.6 TB/s)
// M <= 64 (one 64-row slab): the round-5 constants.  On the grid of profiles/rb8_grid_r06_small.jsonl (M = 40 / 48 / 64, 48 cells) their pick
// is within 1 % of the per-cell best; the set above, fitted to two and more slabs, is 7 % off there (it prices a meeting too high for a
// single slab's parts, which arrive together)
constexpr Rb8Cost kRb8Cost64One{5.5, {0.333, 0.35, 0.52}, 3.17, 4.53, 0.4};
constexpr double kRb8StreamBytesPerUsOne = 5.85e6;
inline Rb8Plan rb8_plan(int64_t M, int64_t N, int64_t K, int bm_forced = 0) {
  const int64_t ksteps = K >> 7;
  const bool one = M <= 64;
  const double hbm_us = (double)N * (double)K / (one ? kRb8StreamBytesPerUsOne : kRb8StreamBytesPerUs);
  Rb8Plan best{M <= 64 ? 64 : 128, 128, 1};
  double best_t = 1e30;
  for (int bm : {128, 64}) {
    // (product: 64 rows up to 64, either from 65 to 512 -- the grid's range -- and 128 beyond)
    if (bm_forced ? bm != bm_forced : ((bm == 128 && M <= 64) || (bm == 64 && M > 512))) continue;
    const Rb8Cost& c = (bm == 64) ? (one ? kRb8Cost64One : kRb8Cost64) : kRb8Cost128;
    const int64_t slabs = (M + bm - 1) / bm;
    for (int bn : {128, 64, 32}) {
      const int64_t tiles = ((N + bn - 1) / bn) * slabs;
      const int64_t fit = (int64_t)kSplitMaxTiles * 128 * 128 / (tiles * bn * bm) * 4 / 5;  // (x 4 / 5: the two-level meeting parks S + S / 4 tiles)
      const double step = c.step[bn == 32 ? 0 : bn == 64 ? 1 : 2];
      for (int S : {1, 2, 3, 4, 6, 8}) {
        if (S > 1 && (S > fit || S > std::max<int64_t>(1, ksteps / 4))) continue;
        const int64_t wgs = tiles * S, rounds = (wgs + 255) / 256, steps = (ksteps + S - 1) / S;
        if (S > 1 && wgs > 256) continue;  // K is never cut into a second round of the chip (1280 x 8192 at M = 2048: 3 parts 52 us, one 45)
        const double loop = std::max((double)steps * step * (double)rounds, hbm_us);
        const double meet = S == 1 ? 0.0 : (S <= 4 ? c.meet1 : c.meet2) + c.gather * (S - 1) * bn / 128.0 * (bm / 128.0);
        const double t = c.fixed * (double)rounds + loop + meet;
        if (t < best_t) { best_t = t; best = Rb8Plan{bm, bn, S}; }
      }
    }
  }
  return best;
}

}  // namespace
// what rb8_run picks without tuning overrides (host logic only: ao_gemm8_plan, tests/test_host_dispatch.py)
void rb8_plan_query(int64_t M, int64_t N, int64_t K, int* bm, int* bn, int* split) {
  const Rb8Plan plan = rb8_plan(M, N, K);
  const int64_t base = ((N + plan.bn - 1) / plan.bn) * ((M + plan.bm - 1) / plan.bm);
  const int64_t fit = (int64_t)kSplitMaxTiles * 128 * 128 / (base * plan.bn * plan.bm) * 4 / 5;
  *bm = plan.bm;
  *bn = plan.bn;
  *split = (int)std::max<int64_t>(1, std::min<int64_t>(plan.split, fit));
}
namespace {

template <int KIND>
int rb8_run(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y, int64_t M,
            int64_t N, int64_t K, hipStream_t stream) {
  Rb8Args p{};
  p.a = a; p.b = b; p.scale_a = scale_a; p.scale_b = scale_b; p.bias = bias; p.y = y;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  // slab height, tile width, K parts: the cost model's pick (64-row slabs for M <= 64 always; tuning key 3 forces a height)
  Rb8Plan plan = rb8_plan(M, N, K, (g_rb8_bm == 64 || g_rb8_bm == 128) ? g_rb8_bm : 0);
  const int bm = plan.bm;
  const int64_t slabs = (M + bm - 1) / bm, ksteps = K >> 7;
  if (g_fp8_rb_force == 3) plan.bn = 64;
  int bn = (g_rb8_bn == 32 || g_rb8_bn == 64 || g_rb8_bn == 128) ? g_rb8_bn : plan.bn;
  const int64_t base = ((N + bn - 1) / bn) * slabs;
  const int64_t fit = (int64_t)kSplitMaxTiles * 128 * 128 / (base * bn * bm) * 4 / 5;
  int split = (bn == plan.bn) ? plan.split : (int)std::max<int64_t>(1, std::min<int64_t>({256 / base, fit, 16, ksteps / 4}));
  if (g_rb8_split > 0) split = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)g_rb8_split, fit, 16, ksteps}));
  split = (int)std::max<int64_t>(1, std::min<int64_t>(split, fit));
  if (bn == 32) return (bm == 64) ? launch_rb8<2, KIND, 4>(p, split, stream) : launch_rb8<2, KIND, 8>(p, split, stream);
  if (bm == 64) return bn == 64 ? launch_rb8<4, KIND, 4>(p, split, stream) : launch_rb8<8, KIND, 4>(p, split, stream);
  return bn == 64 ? launch_rb8<4, KIND, 8>(p, split, stream) : launch_rb8<8, KIND, 8>(p, split, stream);
}

}  // namespace

int fp8_rowwise_rb(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                   int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  return rb8_run<RB8_FP8>(a, b, scale_a, scale_b, bias, y, M, N, K, stream);
}

int int8_scaled_rb(const int8_t* a, const int8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                   int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  return rb8_run<RB8_INT8>(reinterpret_cast<const uint8_t*>(a), reinterpret_cast<const uint8_t*>(b), scale_a, scale_b, bias, y, M, N, K, stream);
}

// MXFP8 grouped GEMM (aten::_scaled_grouped_mm as called from mxfp8_grouped_mm.py:541, numerics of :959-1023): out rows of
// group e = dq(a rows) . dq(b[e])^T with the E8M0 block scales as MFMA operands.  Group sizes live on the device:
// `rows_hint` = rows the largest group is expected to have; the grid provides ceil(rows_hint / 128) slabs per group
// (callers pass M_total when they cannot bound it: empty slabs exit at once).
int mxfp8_grouped_rb(const uint8_t* a, const uint8_t* a_scale, const uint8_t* b, const uint8_t* b_scale, const int32_t* offs, uint16_t* out,
                     int64_t M_total, int64_t N, int64_t K, int64_t E, int64_t rows_hint, hipStream_t stream) {
  Rb8Args p{};
  p.a = a; p.b = b; p.y = out; p.a_mx = a_scale; p.b_mx = b_scale; p.offs = offs;
  p.M = (int)M_total; p.N = (int)N; p.K = (int)K; p.E = (int)E;
  // Slab capacity from the average group size (the sizes themselves live on the device): 64 rows for decode-size groups (two
  // workgroups per CU; a group's real m-tile count is found on the device), else 128.
  const int64_t groups = (offs != nullptr ? E : 1);
  const int bm = (g_rb8_bm == 64 || g_rb8_bm == 128) ? g_rb8_bm : (M_total <= 48 * groups) ? 64 : 128;  // (tuning key 3 forces a height: A/B)
  p.slabs = (int)std::max<int64_t>(1, (std::min(rows_hint, M_total) + bm - 1) / bm);
  // 64-column tiles when 128-column ones would not give every CU a workgroup even if every group had tokens
  // (cutting K into 2 - 4 parts that meet through the split-K workspace -- finer work items for the last round when few experts
  // have tokens -- was measured and dropped: w1 64.6 -> 76.7 us, w2 78 -> 79 us with three experts hit, 105 -> 113 us with all
  // eight: priming a part's rings costs what the shorter tail saves)
  // decode-size groups: the stream-K form when its bounds hold (group table in registers: E <= 64; 32-bit step counter; a
  // ticket per tile).  Mixtral's shapes, hipGraph, us (w1 / w2; profiles/mx_rb_trace_r03.txt session G): three experts hit 58.0 / 71.3
  // with one workgroup per tile -> 56.1 / 52.2; all eight 111 / 110 -> 101 / 91.
  if (bm == 64 && g_mx_stream != 0 && groups <= 64) {
    const int64_t tiles = ((M_total + 63) / 64 + groups) * ((N + 63) / 64);  // (64-column tiles: the bound of every form)
    if (tiles <= kSplitMaxTickets && tiles * (K >> 7) < (1ll << 31))
    {
      // scales fetched per 4 steps when K allows (16-byte pieces of 16-byte-aligned scale rows), else per step
      const bool quad = g_mx_quad && K % 512 == 0 && ((uintptr_t)a_scale % 16 == 0) && ((uintptr_t)b_scale % 16 == 0);
      // round 6: ONE 16-wave workgroup per CU over 256-column tiles where the scales can be fetched per 4 steps; other K (and variant 129,
      // which tests that form on every shape) the 8-wave form of rounds 3 - 5 with the scales fetched per step, two workgroups per CU
      if (quad) return launch_mx_stream<16, 3, 4>(p, stream);
      return launch_mx_stream<8, 3, 1>(p, stream);
    }
  }
  // larger groups (and more than 64 experts): one workgroup per (slab, tile); scales per 4 steps when K allows
  const bool quad = g_mx_quad && K % 512 == 0 && ((uintptr_t)a_scale % 16 == 0) && ((uintptr_t)b_scale % 16 == 0);
  if (bm == 64) {
    return quad ? launch_rb8<4, RB8_MX, 4, false, 4>(p, 1, stream) : launch_rb8<4, RB8_MX, 4, true>(p, 1, stream);
  }
  if (((N + 127) / 128) * groups * p.slabs < 400) return quad ? launch_rb8<4, RB8_MX, 8, false, 4>(p, 1, stream) : launch_rb8<4, RB8_MX, 8>(p, 1, stream);
  return quad ? launch_rb8<8, RB8_MX, 8, false, 4>(p, 1, stream) : launch_rb8<8, RB8_MX, 8>(p, 1, stream);
}

// The same with the activations' 1 x 32 cast fused into the A-fill (reference call order mxfp8_grouped_mm.py:330-371: to_mx(A)
// then the grouped mm): `a` is the BF16 [M_total, K] matrix.  Decode-size groups on the 16-wave stream-K kernel only: mx_dyn_fits says
// whether a shape is taken (callers cast + multiply otherwise).
// (products: 1, or 2 for the pair forms -- two weight tensors of one shape against the same activations in one launch)
bool mxfp8_grouped_dyn_fits(int64_t M_total, int64_t N, int64_t K, int64_t E, bool have_offs, int products) {
  const int64_t groups = have_offs ? E : 1;
  if (M_total <= 0 || M_total > 48 * groups || groups > 64 || K % 512 != 0 || N % 16 != 0) return false;
  if (M_total * K >= (1ll << 31) || N * K >= (1ll << 32)) return false;
  const int64_t tiles = ((M_total + 63) / 64 + groups) * ((N + 63) / 64) * products;
  return tiles <= kSplitMaxTickets && tiles * (K >> 7) < (1ll << 31);
}
// a: BF16 activations (scaling_mode 0 / 1: the cast fused into the A-fill) or, with a_scale != nullptr, their e4m3 codes.  b2 / b2_scale /
// out2: the second product of the pair forms, or null.
int mxfp8_grouped_stream16(const void* a, const uint8_t* a_scale, const uint8_t* b, const uint8_t* b_scale, const uint8_t* b2, const uint8_t* b2_scale,
                           const int32_t* offs, uint16_t* out, uint16_t* out2, int64_t M_total, int64_t N, int64_t K, int64_t E, int scaling_mode,
                           hipStream_t stream) {
  Rb8Args p{};
  p.a = reinterpret_cast<const uint8_t*>(a); p.b = b; p.y = out; p.a_mx = a_scale; p.b_mx = b_scale; p.offs = offs;
  p.b2 = b2; p.b2_mx = b2_scale; p.y2 = out2;
  p.M = (int)M_total; p.N = (int)N; p.K = (int)K; p.E = (int)E;
  if (a_scale != nullptr) return launch_mx_stream<16, 3, 4>(p, stream);
  return scaling_mode == 1 ? launch_mx_stream<16, 3, 4, 2>(p, stream) : launch_mx_stream<16, 3, 4, 1>(p, stream);
}

// Float8Tensor's _grouped_mm, rowwise (float8_tensor.py:1085-1122 -> scaled_grouped_mm with RowWise scales):
//   out[rows of group e] = bf16((a . b[e]^T) * scale_a[m] * scale_b[e][n]); the grouping of the MXFP8 form, the epilogue of fp8 rowwise
int fp8_rowwise_grouped_rb(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, const int32_t* offs, uint16_t* out,
                           int64_t M_total, int64_t N, int64_t K, int64_t E, hipStream_t stream) {
  Rb8Args p{};
  p.a = a; p.b = b; p.scale_a = scale_a; p.scale_b = scale_b; p.y = out; p.offs = offs;
  p.M = (int)M_total; p.N = (int)N; p.K = (int)K; p.E = (int)E;
  const int bm = (g_rb8_bm == 64 || g_rb8_bm == 128) ? g_rb8_bm : (M_total <= 48 * E) ? 64 : 128;
  p.slabs = (int)std::max<int64_t>(1, (M_total + bm - 1) / bm);
  if (bm == 64) return launch_rb8<4, RB8_FP8_GROUPED, 4>(p, 1, stream);
  return (((N + 127) / 128) * E * p.slabs < 400) ? launch_rb8<4, RB8_FP8_GROUPED, 8>(p, 1, stream) : launch_rb8<8, RB8_FP8_GROUPED, 8>(p, 1, stream);
}

}  // namespace ao
