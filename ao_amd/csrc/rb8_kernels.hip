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
#include "splitk.h"

#include <algorithm>

namespace ao {
namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int kStages = 3;   // activation ring (shared), filled 2 steps ahead
constexpr int kWStages = 6;  // weight ring (per wave), filled 5 steps ahead: the HBM stream needs the bytes in flight
constexpr int kABuf = 128 * 128;  // one activation stage: 128 rows x 128 k bytes

struct Rb8Args {
  const uint8_t* a;       // [M][K] e4m3 / int8
  const uint8_t* b;       // [N][K] e4m3 / int8
  const float* scale_a;   // [M]
  const float* scale_b;   // [N]
  const uint16_t* bias;   // [N] bf16 or null
  uint16_t* y;            // [M][N] bf16
  int M, N, K;
  float* ws;
  unsigned* tickets;
  unsigned long long* trace;  // profiling build only
};

// TRACE (profiling build): s_memtime stamps of wave 0, 16 u64 per workgroup: entry, ring primed, barrier of steps 0..7 passed,
// loop done, meeting done, exit
template <int WAVES, bool INT8, bool TRACE = false>
__global__ __launch_bounds__(64 * WAVES) void rb8_kernel(Rb8Args p) {
  unsigned long long ts[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (TRACE) ts[0] = __builtin_amdgcn_s_memtime();
  constexpr int ADMA = 16 / WAVES;  // activation DMAs per wave and stage (8 rows each)
  constexpr int LPS = ADMA + 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][128][128 B] a | [WAVES][6][2 KiB] b

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 128;
  const int ntiles = p.N >> 4;
  const int tile = blockIdx.x * WAVES + wave;
  const int tile_c = min(tile, ntiles - 1);  // tiles past N alias the last one; never stored
  const int ksteps = p.K >> 7;
  const int S = gridDim.z, ks = blockIdx.z;
  const int k0 = (int)(((long long)ksteps * ks) / S);
  const int nk = (int)(((long long)ksteps * (ks + 1)) / S) - k0;

  uint32_t aoff[ADMA];
#pragma unroll
  for (int i = 0; i < ADMA; ++i) {
    const int row = 8 * (ADMA * wave + i) + (lane >> 3);
    aoff[i] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)p.K + ((((lane & 7) ^ (row >> 1)) & 7) << 4);
  }
  // weight DMA i (0, 1) of a step fetches rows 8 i + (lane >> 3) of the n-tile as FULL 128-byte lines (chunk position lane & 7,
  // same swizzle as the activations); half-line requests -- one lane group per 64 bytes -- ran the stream at 3.7 TB/s
  uint32_t boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * i + (lane >> 3);
    boff[i] = (uint32_t)row * (uint32_t)p.K + ((((lane & 7) ^ (row >> 1)) & 7) << 4);
  }
  const uint8_t* brows = p.b + (size_t)tile_c * 16 * p.K;
  const uint32_t a_lds = lds_offset(smem);
  const uint32_t w_lds = a_lds + kStages * kABuf + wave * (kWStages * 2048);
  // k clamped: the fills past the end re-read the last step (unused)
  auto issue_a = [&](int stage, int k) {
    const int kk = k0 + min(k, nk - 1);
#pragma unroll
    for (int i = 0; i < ADMA; ++i) dma_b128_s(p.a + (size_t)kk * 128, aoff[i], a_lds + stage * kABuf + (ADMA * wave + i) * 1024);
  };
  auto issue_w = [&](int stage, int k) {
    const int kk = k0 + min(k, nk - 1);
    dma_b128_nt_s(brows + (size_t)kk * 128, boff[0], w_lds + stage * 2048);
    dma_b128_nt_s(brows + (size_t)kk * 128, boff[1], w_lds + stage * 2048 + 1024);
  };

  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // A operand of lane (row r = nl, kq): chunks kq and 4 + kq of the row, at positions chunk ^ ((r >> 1) & 7)
  const int pa = nl * 128 + (((kq ^ (nl >> 1)) & 7) << 4);  // second half: ^ 64; + 2048 per m-tile

  // Issue order is w(0..2) | a(0) w(3) | a(1) w(4), then per step a(k+2) w(k+5): when step k starts, the LPS + 2 youngest
  // requests are a(k+1), w(k+4) and w(k+3); everything older -- a(k) and w(k) .. w(k+2) -- has landed.
  issue_w(0, 0); issue_w(1, 1); issue_w(2, 2);
  issue_a(0, 0); issue_w(3, 3);
  issue_a(1, 1); issue_w(4, 4);
  if (TRACE) ts[1] = __builtin_amdgcn_s_memtime();
  int stage = 0, wstage = 0;
  for (int k = 0; k < nk; ++k) {
    wait_vmcnt<LPS + 2>();
    // everyone's share of the activation tile has landed, and everyone has finished reading step k - 1
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (TRACE && k < 8) ts[2 + k] = __builtin_amdgcn_s_memtime();
    issue_a((stage == 0) ? 2 : stage - 1, k + 2);
    issue_w((wstage == 0) ? kWStages - 1 : wstage - 1, k + 5);
    const char* A = smem + stage * kABuf;
    const char* W = smem + kStages * kABuf + (wave * kWStages + wstage) * 2048;
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(W + pa);  // the n-tile's 16 rows are laid out like an m-tile
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(W + (pa ^ 64));
    const i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(A + mt * 2048 + pa);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(A + mt * 2048 + (pa ^ 64));
      if constexpr (INT8) {  // acc holds int32 bit patterns
        i32x4 c = __builtin_bit_cast(i32x4, acc[mt]);
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a1), __builtin_bit_cast(i32x4, b1), c, 0, 0, 0);
        acc[mt] = __builtin_bit_cast(f32x4, c);
      } else {
        const i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        acc[mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc[mt], 0, 0, 0, 127, 0, 127);
      }
    }
    stage = (stage == 2) ? 0 : stage + 1;
    wstage = (wstage == kWStages - 1) ? 0 : wstage + 1;
  }
  wait_vmcnt<0>();  // the clamped fills past the end still write LDS
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  if (TRACE) ts[10] = __builtin_amdgcn_s_memtime();
  auto dump = [&] {
    if (TRACE && p.trace != nullptr && tid == 0) {
      ts[12] = __builtin_amdgcn_s_memtime();
      unsigned long long* t = p.trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
      for (int i = 0; i < 13; ++i) t[i] = ts[i];
    }
  };
  if (S > 1 && !split_k_meet<8, 64 * WAVES, INT8>(acc, p.ws, p.tickets, blockIdx.y * gridDim.x + blockIdx.x, S, ks, tid, reinterpret_cast<int*>(smem))) {
    dump();
    return;
  }
  if (TRACE) ts[11] = __builtin_amdgcn_s_memtime();

  // D layout: lane (col = nl, kq) holds rows 4 kq + {0..3} of each 16 x 16 tile
  if (tile >= ntiles) { dump(); return; }
  const int n = tile * 16 + nl;
  const float* __restrict__ scale_a = p.scale_a;
  uint16_t* __restrict__ y = p.y;
  const float sb = p.scale_b[n];
  const float bias = p.bias != nullptr ? bf16_lo_to_f32(p.bias[n]) : 0.f;
  float sa[32];  // all row scales first: the stores below must not sit between dependent loads
#pragma unroll
  for (int i = 0; i < 32; ++i) sa[i] = scale_a[min(m0 + (i >> 2) * 16 + kq * 4 + (i & 3), p.M - 1)];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
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
  dump();
}

unsigned long long* g_fp8_rb_trace = nullptr;  // profiling only (ao_int4_set_trace shares the pointer)

template <int WAVES, bool INT8>
int launch_rb8(Rb8Args p, int split, hipStream_t stream) {
  constexpr int BN = WAVES * 16;
  dim3 grid((unsigned)((p.N + BN - 1) / BN), (unsigned)((p.M + 127) / 128), (unsigned)split), block(64 * WAVES);
  constexpr size_t smem = (size_t)kStages * kABuf + (size_t)WAVES * kWStages * 2048;
  if (split > 1) {
    AO_REQUIRE((int64_t)grid.x * grid.y * split * BN <= (int64_t)kSplitMaxTiles * 128, "rb8: %u x %u tiles x %d parts exceed the split-K workspace",
               grid.x, grid.y, split);
    if (int rc = splitk_workspace(&p.ws, &p.tickets)) return rc;
  }
  p.trace = g_fp8_rb_trace;
  auto kern = (p.trace != nullptr) ? rb8_kernel<WAVES, INT8, true> : rb8_kernel<WAVES, INT8, false>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[p.trace != nullptr]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_failed(e, "hipFuncSetAttribute(rb8_kernel)");
    attr_set[p.trace != nullptr] = true;
  }
  ao::launch(kern, grid, block, smem, stream, p);
  AO_LAUNCH_CHECK("rb8_kernel launch");
  return AO_OK;
}

int g_fp8_rb_force = 0;  // profiling only: 0 product heuristic, 1 never, 2 always, 3 always + 64-column tiles, two workgroups per CU

}  // namespace

void fp8_rowwise_rb_set_mode(int mode) { g_fp8_rb_force = mode; }
void fp8_rowwise_rb_set_trace(unsigned long long* p) { g_fp8_rb_trace = p; }
bool fp8_rowwise_rb_forced() { return g_fp8_rb_force >= 2; }

// True when this kernel is the better choice: the 128 x 128 GEMM grid would leave most of the chip idle (same rule for int8).
bool fp8_rowwise_rb_preferred(int64_t M, int64_t N, int64_t K) {
  if (K % 128 != 0 || N % 16 != 0 || M * K >= (1ll << 32) || N * K >= (1ll << 32)) return false;
  if (g_fp8_rb_force == 1) return false;
  if (g_fp8_rb_force >= 2) return true;
  return ((N + 127) / 128) * ((M + 127) / 128) < 190;
}

namespace {

template <bool INT8>
int rb8_run(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y, int64_t M,
            int64_t N, int64_t K, hipStream_t stream) {
  Rb8Args p{a, b, scale_a, scale_b, bias, y, (int)M, (int)N, (int)K, nullptr, nullptr, nullptr};
  const int64_t slabs = (M + 127) / 128, ksteps = K >> 7;
  // 128-column tiles while they give ~half a chip of workgroups before splitting, else 64-column tiles; K cut into at most
  // 16 parts of >= 4 steps so that the grid approaches one workgroup per CU
  const bool wide = ((N + 127) / 128) * slabs * std::min<int64_t>(16, std::max<int64_t>(1, ksteps / 4)) >= 190;
  const bool narrow = !wide || g_fp8_rb_force == 3;
  const int bn = narrow ? 64 : 128;
  const int64_t base = ((N + bn - 1) / bn) * slabs;
  const int64_t fit = (int64_t)kSplitMaxTiles * 128 / (base * bn);
  const int64_t target = (g_fp8_rb_force == 3) ? 512 : 256;
  const int split = (int)std::max<int64_t>(1, std::min<int64_t>({target / base, fit, 16, ksteps / 4}));
  return narrow ? launch_rb8<4, INT8>(p, split, stream) : launch_rb8<8, INT8>(p, split, stream);
}

}  // namespace

int fp8_rowwise_rb(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                   int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  return rb8_run<false>(a, b, scale_a, scale_b, bias, y, M, N, K, stream);
}

int int8_scaled_rb(const int8_t* a, const int8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                   int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  return rb8_run<true>(reinterpret_cast<const uint8_t*>(a), reinterpret_cast<const uint8_t*>(b), scale_a, scale_b, bias, y, M, N, K, stream);
}

}  // namespace ao
