// 8-bit linears at 16 < M <= 256 with few output tiles (round 4): the TP-shard / batched-decode sizes of BASELINE config 4, where the
// op is bound by reading the weights once and by the fixed costs of a short launch.
//
//   fp8 :  y[M,N] = bf16((a . b^T) * scale_a[m] * scale_b[n] + bias[n])     aten::_scaled_mm rowwise, float8/inference.py:104-123
//   int8:  y[M,N] = bf16(bf16(i32(a . b^T) * sx[m]) * sw[n] + bias[n])      _int_mm + scales, int8_tensor.py:305-359
//   fused: a, scale_a = per-row cast of x (bf16) inside the same launch     float8_tensor.py:347-355, int8_tensor.py:287-294
//
// rb8_kernel (round 1-3) stages BOTH operands through LDS-DMA rings: 144 KiB of LDS, ONE workgroup per CU, so ring priming (2-3 us),
// the split-K meeting and the epilogue of a 10-20 us launch overlap with nothing.  Here the weights take dec8_kernel's path instead --
// full 128-byte lines straight into a static REGISTER ring (G steps = G x 2 KiB per wave in flight, compiler-counted waits), turned
// into the MFMA operand layout through a wave-private 2.25 KiB slab -- and only the activation tile goes through a shared,
// double-buffered LDS tile (rows 144 bytes apart: conflict-free fragment reads): 54 KiB per workgroup, TWO workgroups per CU; one's
// fixed costs hide behind the other's k loop, and twice the weight bytes are in flight per CU.
//   * a workgroup = 8 waves = 8 n-tiles (128 columns) x MT m-tiles (32 / 64 / 128 rows) x one K part of nk steps (nk % G == 0);
//   * per step and wave: 2 weight loads + NA activation loads (issued G resp. PA steps ahead), 2 + NA ds_write_b128, 2 + 2 MT
//     ds_read_b128, MT (fp8) or 2 MT (int8) MFMAs, ONE LDS-only barrier;
//   * K parts meet through the two-level meeting of splitk.h (groups of four parts: two dependent round trips instead of S / 4);
//   * the dynamic-activation entry points (ao_*_dynamic_linear at 16 < M <= 256) run the stand-alone per-row cast into a scratch
//     area of the stream's split-K workspace and then this kernel on it: two launches behind one call.  Round 4 built the cast INTO the
//     kernel (rows shared out through a ticket counter, a grid-wide wait before the k loop): it measured 15 - 20 us slower per linear
//     than cast + matmul (profiles/mid8_sweep_r04.txt) and its ticket / time-out paths were the round-4 advisor's two
//     findings; round 5 removed it.
#include "common.h"
#include "quant_math.h"
#include "splitk.h"

#include <algorithm>
#include <type_traits>

namespace ao {
thread_local int g_mid8_mode = 0;  // ao_gemm8_set_variant 300: never this kernel; 301: always (where the shape allows); 31S: force S K-parts
namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct Mid8Args {
  const uint8_t* a;        // [M][K] codes
  const uint8_t* b;        // [N][K]
  const float* scale_a;    // [M]
  const float* scale_b;    // [N]
  const uint16_t* bias;    // [N] bf16 or null
  uint16_t* y;             // [M][N] bf16
  int M, N, K;
  float* ws;               // split-K parts
  unsigned* tickets;
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kRow = 144;          // LDS bytes per 128-byte row (weights slab and activation tile): 16 rows hit 64 distinct banks
constexpr int kSlab = 16 * kRow;   // 2304 B per wave
constexpr int kPA = 2;             // activation steps requested ahead (L2 hits)


template <bool INT8, int MT, int G>
__global__ __launch_bounds__(512, 4) void mid8_kernel(Mid8Args p) {
  constexpr int BM = 16 * MT;
  constexpr int AROWS = (BM < 64) ? 64 : BM;  // rows of the LDS activation tile (512 threads fetch 64 rows per pass)
  constexpr int NA = AROWS / 64;              // activation loads per thread and step
  static_assert(G % kPA == 0 && G % 2 == 0, "ring slots and LDS buffers are compile-time indices of the unrolled group");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2][AROWS][144] activation tile | [8][16][144] weight slabs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  char* abuf = smem;
  char* slab = smem + 2 * AROWS * kRow + wave * kSlab;

  const int ntiles = p.N >> 4;
  const int tile = blockIdx.x * 8 + wave;
  const int tile_c = min(tile, ntiles - 1);  // tiles past N alias the last one; never stored
  const int m0 = blockIdx.y * BM;
  const int ksteps = p.K >> 7;
  const int S = gridDim.z, ks = blockIdx.z;
  const int nk = ksteps / S;  // host: ksteps % (S * G) == 0
  const int k0 = ks * nk;

  // ---- weight ring: lane l fetches row (l >> 3) of the tile's rows 0..7 / 8..15, chunk l & 7 of the step's 128 bytes (full lines).
  // Buffer addressing (SGPR descriptor + one 32-bit VGPR offset + an SGPR offset per step): no 64-bit pointers in VGPRs.
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.b), 0, (int)((size_t)p.N * p.K), 0x00020000);
  const uint32_t woff = ((uint32_t)tile_c * 16 + (lane >> 3)) * (uint32_t)p.K + (lane & 7) * 16;
  const uint32_t whalf = 8u * (uint32_t)p.K;
  struct WStage {
    u32x4 b0, b1;
  };
  WStage wr[G];
  auto issue_w = [&](WStage& s, int k) {
    const uint32_t so = (uint32_t)(k0 + min(k, nk - 1)) * 128u;  // past the end: the last step again, unused
    s.b0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)woff, (int)so, 2 /* nt: streamed once */));
    s.b1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)woff, (int)(so + whalf), 2));
  };
#pragma unroll
  for (int g = 0; g < G; ++g) issue_w(wr[g], g);
  __builtin_amdgcn_sched_barrier(0);

  // ---- activation tile: thread t fetches chunk t & 7 of rows (t >> 3) + 64 i (clamped into the matrix: rows past M are never stored)
  uint32_t aoff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) aoff[i] = (uint32_t)min(m0 + (tid >> 3) + 64 * i, p.M - 1) * (uint32_t)p.K + (tid & 7) * 16;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.a), 0, (int)((size_t)p.M * p.K), 0x00020000);
  char* awr = abuf + (tid >> 3) * kRow + (tid & 7) * 16;  // + 64 i rows, + buffer
  u32x4 ar[kPA][NA];
  auto load_a = [&](u32x4 (&dst)[NA], int k) {
    const uint32_t so = (uint32_t)(k0 + min(k, nk - 1)) * 128u;
#pragma unroll
    for (int i = 0; i < NA; ++i) dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)aoff[i], (int)so, 0));
  };
  auto store_a = [&](const u32x4 (&src)[NA], int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(awr + (buf * AROWS + 64 * i) * kRow) = src[i];
  };
  {
    u32x4 first[NA];
    load_a(first, 0);
#pragma unroll
    for (int s = 1; s <= kPA; ++s) load_a(ar[s % kPA], s);
    store_a(first, 0);
  }
  lds_barrier();

  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  char* wwr = slab + (lane >> 3) * kRow + (lane & 7) * 16;  // this lane's piece of slab rows 0..7; rows 8..15: + 8 rows
  const char* wrd = slab + nl * kRow + kq * 16;
  const char* ard = abuf + nl * kRow + kq * 16;

  for (int kb = 0; kb < nk; kb += G) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k = kb + g;
      // A(k + 1) -> the other LDS buffer (its readers finished at the previous barrier), its registers refilled with A(k + 1 + PA)
      store_a(ar[(g + 1) % kPA], (g + 1) & 1);
      load_a(ar[(g + 1) % kPA], k + 1 + kPA);
      // W(k): registers -> slab -> operand layout; the slot refilled with W(k + G)
      *reinterpret_cast<u32x4*>(wwr) = wr[g].b0;
      *reinterpret_cast<u32x4*>(wwr + 8 * kRow) = wr[g].b1;
      issue_w(wr[g], k + G);
      const u32x4 b0 = *reinterpret_cast<const u32x4*>(wrd);
      const u32x4 b1 = *reinterpret_cast<const u32x4*>(wrd + 64);
      const char* A = ard + (g & 1) * AROWS * kRow;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(A + mt * 16 * kRow);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(A + mt * 16 * kRow + 64);
        if constexpr (INT8) {  // acc holds int32 bit patterns
          i32x4 c = __builtin_bit_cast(i32x4, acc[mt]);
          c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, b1), c, 0, 0, 0);
          acc[mt] = __builtin_bit_cast(f32x4, c);
        } else {
          const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
          const i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
          acc[mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc[mt], 0, 0, 0, 127, 0, 127);
        }
      }
      // (the fence keeps a step's MFMAs inside the step: left alone the scheduler carries accumulators across the barrier in renamed
      // registers, and the fused MT = 8 forms spilled their weight ring at the 128-VGPR budget of four waves per SIMD)
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();  // A(k + 1) is in LDS for everyone; everyone is done reading A(k)
    }
  }

  // ---- K parts meet (two-level, part order: reproducible); the last arriver of a tile goes on to the epilogue
  if (S > 1 && !split_k_meet2<MT, 512, INT8, (MT >= 8 ? 2 : 4)>(acc, p.ws, p.tickets, blockIdx.y * gridDim.x + blockIdx.x, S, ks, tid, reinterpret_cast<int*>(smem))) return;
  if (tile >= ntiles) return;

  // D layout: lane (col = nl, kq) holds rows 4 kq + {0..3} of each 16 x 16 tile
  const int n = tile * 16 + nl;
  uint16_t* __restrict__ y = p.y;
  const float* __restrict__ scale_a = p.scale_a;
  const float sb = p.scale_b[n];
  const float bias = p.bias != nullptr ? bf16_lo_to_f32(p.bias[n]) : 0.f;
  float sa[4 * MT];  // all row scales first: the stores below must not sit between dependent loads
#pragma unroll
  for (int i = 0; i < 4 * MT; ++i) {
    const int m = min(m0 + (i >> 2) * 16 + kq * 4 + (i & 3), p.M - 1);
    sa[i] = scale_a[m];
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + mt * 16 + kq * 4 + r;
      if (m < p.M) {
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

constexpr int kG = 4;  // ring depth of the k loop (G x 2 KiB per wave in flight; 16 waves per CU)

struct Mid8Plan {
  int mt, split;
};

// Rows: 32 / 64 / 128-row slabs.  K parts: ~one workgroup per CU, parts that divide the K steps evenly.
// Product dispatch (cold weights, 70B / TP8 fp8 shards, profiles/mid8_sweep_r04.txt): the kernel beats the round-3 paths at
// 17 .. 32 rows on long K (qkv 1280 x 8192 at M = 32: 11.1 us against 15.1; gate_up 7168 x 8192: 17.2 against 27.8) and loses
// everywhere else (M = 128: qkv 21 us against rb8_kernel's 16, gate_up 28 against 26.5, down 27 against 19 -- its k loop waits on
// activation / weight requests only 2 / 4 steps old and on every A-fragment read, at the 128 VGPRs of four waves per SIMD), so that
// is all it takes by itself; 301 / 31S force it anywhere for A/B runs and the parity tests.
bool mid8_plan(int64_t M, int64_t N, int64_t K, Mid8Plan* out) {
  if (g_mid8_mode == 300) return false;
  if (M <= 16 || M > 256 || N % 16 != 0 || K % (128 * kG) != 0 || M * K >= (1ll << 31) || N * K >= (1ll << 31)) return false;
  const bool forced = g_mid8_mode == 301 || (g_mid8_mode >= 310 && g_mid8_mode < 330);
  if (!forced && !(M <= 32 && K >= 4096)) return false;
  const int mt = (M <= 32) ? 2 : (M <= 64) ? 4 : 8;
  const int64_t slabs = (M + 16 * mt - 1) / (16 * mt), cols = (N + 127) / 128, groups = K / (128 * kG);
  const int64_t base = cols * slabs;
  if (!forced && base >= 400) return false;  // enough tiles for the tiled GEMMs
  int64_t want = std::max<int64_t>(1, 256 / base);
  if (g_mid8_mode >= 310 && g_mid8_mode < 330) want = g_mid8_mode - 310;
  int split = 1;
  for (int64_t s = 1; s <= std::min<int64_t>(groups, 16); ++s)
    if (groups % s == 0 && s <= want) split = (int)s;
  // the meeting's workspace: (S + ceil(S / 4)) parked tiles per output tile
  // Round 6 (profiles/other_shapes_forms_r06.jsonl): a K whose 512-byte groups do not factor (K = 18944: 37 groups, prime -- one part where 9
  // were wanted) leaves a few dozen workgroups on the chip: down 3584 x 18944 at M = 24 / 32 took 68 us here against 23 through rb8_kernel.
  // Refused when the divisors of the group count give less than half the parts the shape would otherwise get.
  if (!forced && 2 * split < std::min<int64_t>({want, groups, 16})) return false;
  // ... and where they leave about half of the chip without a workgroup (8192 x 7168: 64 tiles x 2 parts, 5120 x 13824: 40 x 3) rb8_kernel is
  // 7 - 8 % ahead (20.4 -> 19.0 us, 24.4 -> 22.4); every shape of round 4's fit makes 160 .. 256 workgroups (a single column tile is a test shape)
  if (!forced && base >= 16 && base * split < 144) return false;
  if (split > 1 && base * (split + (split + 3) / 4) * 128 * 16 * mt > (int64_t)kSplitMaxTiles * 128 * 128) return false;
  if (split > 1 && base * (1 + (split + 3) / 4) > kSplitMaxTickets - 8) return false;
  *out = Mid8Plan{mt, split};
  return true;
}

template <bool INT8, int MT>
int launch_mid8(Mid8Args p, int split, hipStream_t stream) {
  constexpr int BM = 16 * MT, AROWS = (BM < 64) ? 64 : BM;
  constexpr size_t smem = (size_t)2 * AROWS * kRow + 8 * kSlab;
  const dim3 grid((unsigned)((p.N + 127) / 128), (unsigned)((p.M + BM - 1) / BM), (unsigned)split), block(512);
  const size_t tiles = (size_t)grid.x * grid.y;
  const size_t part_floats = (split > 1) ? tiles * (split + (split + 3) / 4) * 128 * BM : 0;
  if (split > 1) {
    if (int rc = splitk_workspace(stream, &p.ws, &p.tickets, part_floats, split)) return rc;
  }
  auto kern = mid8_kernel<INT8, MT, kG>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, "hipFuncSetAttribute(mid8_kernel)")) return rc;
  ao::launch(kern, grid, block, smem, stream, p);
  AO_LAUNCH_CHECK("mid8_kernel launch");
  return AO_OK;
}

template <bool INT8>
int run_mid8(const Mid8Args& p, const Mid8Plan& plan, hipStream_t stream) {
  switch (plan.mt) {
    case 2: return launch_mid8<INT8, 2>(p, plan.split, stream);
    case 4: return launch_mid8<INT8, 4>(p, plan.split, stream);
    default: return launch_mid8<INT8, 8>(p, plan.split, stream);
  }
}

}  // namespace

bool mid8_takes(int64_t M, int64_t N, int64_t K) {
  Mid8Plan plan;
  return mid8_plan(M, N, K, &plan);
}
bool mid8_takes_fused(int64_t M, int64_t N, int64_t K) { return K <= 16384 && mid8_takes(M, N, K); }

int mid8_scaled(bool int8, const void* a, const float* scale_a, const void* b, const float* scale_b, const uint16_t* bias, uint16_t* y, int64_t M,
                int64_t N, int64_t K, hipStream_t stream) {
  Mid8Plan plan;
  if (!mid8_plan(M, N, K, &plan)) {
    set_error("mid8_scaled: shape M=%lld N=%lld K=%lld not covered", (long long)M, (long long)N, (long long)K);
    return AO_ERR_INVALID_ARGUMENT;
  }
  Mid8Args p{};
  p.a = static_cast<const uint8_t*>(a); p.b = static_cast<const uint8_t*>(b); p.scale_a = scale_a; p.scale_b = scale_b; p.bias = bias; p.y = y;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  return int8 ? run_mid8<true>(p, plan, stream) : run_mid8<false>(p, plan, stream);
}

// ao_*_dynamic_linear at 16 < M <= 256: the per-row cast (the stand-alone kernel: same arithmetic, same bits) into a scratch area BEHIND
// the parts of the meeting in the stream's split-K workspace, then the kernel above on it -- stream-ordered, two launches.
int mid8_dynamic(bool int8, const uint16_t* x, const void* b, const float* scale_b, const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K,
                 hipStream_t stream) {
  Mid8Plan plan;
  if (K > 16384 || !mid8_plan(M, N, K, &plan)) {
    set_error("mid8_dynamic: shape M=%lld N=%lld K=%lld not covered", (long long)M, (long long)N, (long long)K);
    return AO_ERR_INVALID_ARGUMENT;
  }
  const size_t tiles = (size_t)((N + 127) / 128) * (size_t)((M + 16 * plan.mt - 1) / (16 * plan.mt));
  const size_t part_floats = (plan.split > 1) ? tiles * (plan.split + (plan.split + 3) / 4) * 128 * 16 * plan.mt : 0;
  const size_t code_floats = ((size_t)M * K + 15) / 16 * 4, scale_floats = (size_t)((M + 3) / 4 * 4);
  float* ws = nullptr;
  unsigned* tickets = nullptr;
  if (int rc = splitk_workspace(stream, &ws, &tickets, part_floats + code_floats + scale_floats)) return rc;
  uint8_t* xq = reinterpret_cast<uint8_t*>(ws + part_floats);
  float* xs = ws + part_floats + code_floats;
  if (int rc = int8 ? ao_int8_quantize_rowwise(x, reinterpret_cast<int8_t*>(xq), xs, M, K, stream) : ao_fp8_quantize_rowwise(x, xq, xs, M, K, stream)) return rc;
  return mid8_scaled(int8, xq, xs, b, scale_b, bias, y, M, N, K, stream);
}

}  // namespace ao
