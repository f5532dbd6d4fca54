// Tiled MFMA GEMM for 1-byte operands on gfx950: y = epilogue(A[M,K] @ B[N,K]^T)
//   * int8 x int8 -> int32  (aten::_int_mm / the Int8Tensor linear:
//     torchao/quantization/quantize_/workflows/int8/kernels.py:114-144,
//     int8_tensor.py:305-359) on v_mfma_i32_32x32x32_i8
//   * e4m3 x e4m3 -> fp32   (aten::_scaled_mm rowwise, torchao/float8/inference.py:
//     86-123) on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (the
//     K=64 form is the only full-rate fp8 path on CDNA4)
// Both operands are K-contiguous ("NT"), so A and B tiles are staged the same
// way: 128 x 128-byte tiles, 16 B per thread per pass, through padded LDS rows
// (144 B stride: ds_read_b128 conflict-free), double buffered, one barrier per
// K step.  4 waves as 2x2, each 64x64 = 2x2 MFMA tiles of 32x32.
// The epilogue applies the reference's scaling sequence in registers and stores
// bf16 (or raw int32 for _int_mm).
#include <string>

#include "common.h"

namespace ao {
bool gemm8_p8_fits(int64_t M, int64_t N, int64_t K);  // gemm8_p8_kernels.hip (epi numbering = enum Epilogue)
void gemm8_p8_set_group_rows(int v);
void gemm8_p8_set_split(int v);
void gemm8_p8_set_persistent(int v);
bool gemm8_p8_persistent_shape(int64_t M, int64_t N, int64_t K);  // the persistent form's product rule (shape part; 16-byte-aligned scales assumed)
void gemm8_p8h_set_form(int v);
int gemm8_p8(int epi, const uint8_t* a, const uint8_t* b, const float* row_scale, const float* col_scale, const uint16_t* bias, void* out,
             int64_t M, int64_t N, int64_t K, hipStream_t stream);

// rb8_kernels.hip: weight-streaming kernels for problems with few output tiles
int gemm8_p8h(int epi, const uint8_t* a, const uint8_t* b, const float* row_scale, const float* col_scale, const uint16_t* bias, void* out,
              int64_t M, int64_t N, int64_t K, hipStream_t stream);
bool gemm8_p8h_band(int64_t M, int64_t N, int64_t K);
int gemm8_p8h_parts(int64_t M, int64_t N, int64_t K);
void rb8_plan_query(int64_t M, int64_t N, int64_t K, int* bm, int* bn, int* split);
bool fp8_rowwise_rb_preferred(int64_t M, int64_t N, int64_t K);
bool rb8_small_m_preferred(int64_t M, int64_t N, int64_t K);  // rb8_kernels.hip: 8 .. 64 rows on weights the decode kernels leave
void rb8_set_wave_grid(bool two_by_four);  // rb8_kernels.hip
void rb8_set_tuning(int bn, int split, int ablate);
void rb8_set_slab_rows(int rows);
void fp8_rowwise_rb_set_mode(int mode);
bool fp8_rowwise_rb_forced();
extern thread_local int g_mx_variant;  // stream8_kernels.hip
extern thread_local int g_dec8_mode;   // dec8_kernels.hip
extern thread_local int g_mid8_mode;   // mid8_kernels.hip
bool mid8_takes(int64_t M, int64_t N, int64_t K);
int mid8_scaled(bool int8, const void* a, const float* scale_a, const void* b, const float* scale_b, const uint16_t* bias, uint16_t* y, int64_t M,
                int64_t N, int64_t K, hipStream_t stream);
bool dec8_takes(int64_t M, int64_t N, int64_t K);
int dec8_scaled(bool int8, const void* xq, const float* x_scale, const void* wq, const float* w_scale, const uint16_t* bias, uint16_t* y,
                int64_t M, int64_t N, int64_t K, hipStream_t stream);
void mx_rb_set_stream(int mode, bool quad);
void mx_stream_set_tuning(int proto);
int fp8_rowwise_rb(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                   int64_t M, int64_t N, int64_t K, hipStream_t stream);
int int8_scaled_rb(const int8_t* a, const int8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                   int64_t M, int64_t N, int64_t K, hipStream_t stream);
int int8_scaled_stream(const int8_t* a, const int8_t* b, const float* scale_a, const float* scale_b, const uint16_t* bias, uint16_t* y,
                       int64_t M, int64_t N, int64_t K, hipStream_t stream);
int fp8_rowwise_stream(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b,
                       const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K, hipStream_t stream);

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 128;  // BK in bytes == elements
constexpr int LDS_STRIDE = BK + 16;          // 144 B
constexpr int TILE_BYTES = BM * LDS_STRIDE;  // one operand tile
constexpr int THREADS = 256;

// EPI_FP8_RAW: the unscaled fp32 accumulators (K-sharded row-parallel linears all-reduce them before the scale epilogue)
enum Epilogue { EPI_INT8_SCALED = 0, EPI_INT32 = 1, EPI_FP8_ROWWISE = 2, EPI_FP8_RAW = 3 };

struct Gemm8Args {
  const uint8_t* a;  // [M][K]
  const uint8_t* b;  // [N][K]
  const float* row_scale;  // [M]
  const float* col_scale;  // [N]
  const uint16_t* bias;    // [N] bf16 or null
  void* out;               // bf16 [M][N] or int32 [M][N]
  int M, N, K;
};

template <int EPI>
struct Acc {
  using type = f32x16;
};
template <>
struct Acc<EPI_INT8_SCALED> {
  using type = i32x16;
};
template <>
struct Acc<EPI_INT32> {
  using type = i32x16;
};

template <int EPI>
__global__ __launch_bounds__(THREADS) void gemm8_kernel(Gemm8Args p) {
  constexpr bool IS_INT = (EPI != EPI_FP8_ROWWISE && EPI != EPI_FP8_RAW);
  using acc_t = typename Acc<EPI>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 bufs][A tile | B tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // global -> register staging: 4 passes x 16 B per operand per thread
  // chunk c = pass * 256 + tid: row = c >> 3 (8 chunks per 128-B row), col16 = c & 7
  const int srow = tid >> 3, scol = (tid & 7) * 16;
  const uint8_t* ag[4];
  const uint8_t* bg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = srow + i * 32;
    ag[i] = p.a + (size_t)min(m0 + r, p.M - 1) * p.K + scol;
    bg[i] = p.b + (size_t)min(n0 + r, p.N - 1) * p.K + scol;
  }
  const int ktiles = (p.K + BK - 1) / BK;

  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k = kt * BK + scol;
    const bool in = (k + 16 <= p.K);  // K % 16 == 0: a chunk is entirely in or out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = in ? *reinterpret_cast<const u32x4*>(ag[i] + (size_t)kt * BK) : u32x4{0, 0, 0, 0};
      rb[i] = in ? *reinterpret_cast<const u32x4*>(bg[i] + (size_t)kt * BK) : u32x4{0, 0, 0, 0};
    }
  };
  auto lstore = [&](int buf) {
    char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (srow + i * 32) * LDS_STRIDE + scol;
      *reinterpret_cast<u32x4*>(base + off) = ra[i];
      *reinterpret_cast<u32x4*>(base + TILE_BYTES + off) = rb[i];
    }
  };

  acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  // fragment addressing: row = lane & 31 of the 32-row tile, 32 bytes at (lane >> 5) * 32 of each 64-byte chunk
  const int frow = lane & 31, fk = (lane >> 5) * 32;

  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ktiles) gload(kt + 1);
    const char* abase = smem + buf * 2 * TILE_BYTES + (wm * 64 + frow) * LDS_STRIDE + fk;
    const char* bbase = smem + buf * 2 * TILE_BYTES + TILE_BYTES + (wn * 64 + frow) * LDS_STRIDE + fk;
#pragma unroll
    for (int kc = 0; kc < BK / 64; ++kc) {
      u32x4 af[2][2], bf[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i][0] = *reinterpret_cast<const u32x4*>(abase + i * 32 * LDS_STRIDE + kc * 64);
        af[i][1] = *reinterpret_cast<const u32x4*>(abase + i * 32 * LDS_STRIDE + kc * 64 + 16);
        bf[i][0] = *reinterpret_cast<const u32x4*>(bbase + i * 32 * LDS_STRIDE + kc * 64);
        bf[i][1] = *reinterpret_cast<const u32x4*>(bbase + i * 32 * LDS_STRIDE + kc * 64 + 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (IS_INT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const i32x4 av = {(int)af[i][h].x, (int)af[i][h].y, (int)af[i][h].z, (int)af[i][h].w};
              const i32x4 bv = {(int)bf[j][h].x, (int)bf[j][h].y, (int)bf[j][h].z, (int)bf[j][h].w};
              acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[i][j], 0, 0, 0);
            }
          } else {
            const i32x8 av = {(int)af[i][0].x, (int)af[i][0].y, (int)af[i][0].z, (int)af[i][0].w,
                              (int)af[i][1].x, (int)af[i][1].y, (int)af[i][1].z, (int)af[i][1].w};
            const i32x8 bv = {(int)bf[j][0].x, (int)bf[j][0].y, (int)bf[j][0].z, (int)bf[j][0].w,
                              (int)bf[j][1].x, (int)bf[j][1].y, (int)bf[j][1].z, (int)bf[j][1].w};
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 0, 0, 0, 127, 0, 127);
          }
        }
    }
    if (kt + 1 < ktiles) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = n0 + wn * 64 + j * 32 + (lane & 31);
      if (gn >= p.N) continue;
      float cs = 1.f, bias = 0.f;
      if (EPI != EPI_INT32 && EPI != EPI_FP8_RAW) {
        cs = p.col_scale[gn];
        if (p.bias != nullptr) bias = bf16_lo_to_f32(p.bias[gn]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm >= p.M) continue;
        if (EPI == EPI_INT32) {
          reinterpret_cast<int32_t*>(p.out)[(size_t)gm * p.N + gn] = (int32_t)acc[i][j][r];
        } else if (EPI == EPI_FP8_RAW) {
          reinterpret_cast<float*>(p.out)[(size_t)gm * p.N + gn] = (float)acc[i][j][r];
        } else if (EPI == EPI_INT8_SCALED) {
          // t = bf16(f32(c) * sx[m]);  y = bf16(f32(t) * sw[n] (+ bias))   (int8_tensor.py:315-359)
          const float t = round_bf16((float)acc[i][j][r] * p.row_scale[gm]);
          float y = t * cs;
          if (p.bias != nullptr) y += bias;
          reinterpret_cast<uint16_t*>(p.out)[(size_t)gm * p.N + gn] = f32_to_bf16_bits(y);
        } else {
          float y = (float)acc[i][j][r] * p.row_scale[gm] * cs;
          if (p.bias != nullptr) y += bias;
          reinterpret_cast<uint16_t*>(p.out)[(size_t)gm * p.N + gn] = f32_to_bf16_bits(y);
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// The same GEMM with LDS-DMA staging (global_load_lds_dwordx4: 1 KiB per wave instruction straight
// into LDS, no VGPR round trip, no ds_write pass) -- the single biggest step of the CDNA4 GEMM ladder
// for this two-barrier structure.  LDS-DMA writes lane-linear, so rows cannot be padded; bank
// conflicts of the fragment reads are removed by a swizzle applied on the SOURCE address instead:
// LDS[row][chunk'] holds global 16-byte chunk  chunk' ^ f(row),  f(row) = (row & 7) ^ ((row >> 3) & 7)
// (within the 16 rows of every ds_read_b128 lane group, f is a bijection per row parity -> 16
// distinct 16-byte bank slots).  Needs K % 128 == 0; ragged M/N rows are clamped (never stored).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int swz(int row) { return (row & 7) ^ ((row >> 3) & 7); }

// TM = 32-row MFMA tiles per wave along M: wave tile (TM*32) x 64, workgroup tile (TM*64) x 128.
// At TM = 2 every K step moves 64 KiB from LDS per CU for 512 MFMA cycles per SIMD -- LDS time equals
// MFMA time; TM = 4 (256 x 128 workgroup tile, 128 accumulator registers per lane) does 1024 MFMA
// cycles per 96 KiB of LDS reads.
// WN = waves along n (2: 4-wave workgroup, 128 columns; 4: 8-wave workgroup, 256 columns -- with TM = 4 the
// 256 x 256 tile whose wave tile is 128 x 64 (LDS reads 24 B per MFMA cycle instead of 32) at two waves per SIMD).
// TNJ = 32-column MFMA tiles per wave along N (2 or 4).
template <int EPI, int TM, int WN, int TNJ = 2>
__global__ __launch_bounds__(128 * WN) void gemm8_dma_kernel(Gemm8Args p) {
  constexpr bool IS_INT = (EPI != EPI_FP8_ROWWISE && EPI != EPI_FP8_RAW);
  constexpr int NWAVES = 2 * WN;
  constexpr int WGM = TM * 64;            // workgroup rows
  constexpr int WGN = WN * TNJ * 32;      // workgroup columns
  constexpr int A_TILE = WGM * BK;        // bytes, unpadded
  constexpr int B_TILE = WGN * BK;
  constexpr int A_DMAS = WGM / 8 / NWAVES;  // 1 KiB DMA instructions (8 rows each) per wave for the A tile
  constexpr int B_DMAS = WGN / 8 / NWAVES;
  using acc_t = typename Acc<EPI>::type;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2 bufs][A tile | B tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * WGM, n0 = blockIdx.x * WGN;
  const int ktiles = p.K / BK;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));

  // DMA sources: wave w fills rows (w*A_DMAS + i)*8 .. +7 of the A tile (and (w*B_DMAS + i)*8 of the B tile);
  // lane l lands at row r0 + (l >> 3), position l & 7 -> it must fetch chunk (l & 7) ^ f(row)
  const uint8_t* asrc[A_DMAS];
  const uint8_t* bsrc[B_DMAS];
#pragma unroll
  for (int i = 0; i < A_DMAS; ++i) {
    const int row = (wave * A_DMAS + i) * 8 + (lane >> 3);
    asrc[i] = p.a + (size_t)min(m0 + row, p.M - 1) * p.K + (((lane & 7) ^ swz(row)) << 4);
  }
#pragma unroll
  for (int i = 0; i < B_DMAS; ++i) {
    const int row = (wave * B_DMAS + i) * 8 + (lane >> 3);
    bsrc[i] = p.b + (size_t)min(n0 + row, p.N - 1) * p.K + (((lane & 7) ^ swz(row)) << 4);
  }
  auto stage = [&](int kt, int buf) {
    const uint32_t abase = lds0 + buf * (A_TILE + B_TILE);
#pragma unroll
    for (int i = 0; i < A_DMAS; ++i) dma16(asrc[i] + (size_t)kt * BK, abase + (wave * A_DMAS + i) * 1024);
#pragma unroll
    for (int i = 0; i < B_DMAS; ++i) dma16(bsrc[i] + (size_t)kt * BK, abase + A_TILE + (wave * B_DMAS + i) * 1024);
  };

  acc_t acc[TM][TNJ];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  // fragment addressing: row = lane & 31 of the 32-row MFMA tile, logical chunks (lane >> 5) * 2 + {0, 1} of each
  // 64-byte k-slice; physical chunk = logical ^ f(row)
  const int frow = lane & 31, fc = (lane >> 5) * 2;
  int arow_off[TM], asw[TM], brow_off[TNJ], bsw[TNJ];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ra = wm * (TM * 32) + i * 32 + frow;
    arow_off[i] = ra * BK; asw[i] = swz(ra);
  }
#pragma unroll
  for (int j = 0; j < TNJ; ++j) {
    const int rb = wn * (TNJ * 32) + j * 32 + frow;
    brow_off[j] = rb * BK; bsw[j] = swz(rb);
  }

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ktiles) stage(kt + 1, buf ^ 1);
    const char* abase = smem + buf * (A_TILE + B_TILE);
    const char* bbase = abase + A_TILE;
#pragma unroll
    for (int kc = 0; kc < BK / 64; ++kc) {
      u32x4 af[TM][2], bf[TNJ][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = kc * 4 + fc + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][h] = *reinterpret_cast<const u32x4*>(abase + arow_off[i] + ((c ^ asw[i]) << 4));
#pragma unroll
        for (int j = 0; j < TNJ; ++j) bf[j][h] = *reinterpret_cast<const u32x4*>(bbase + brow_off[j] + ((c ^ bsw[j]) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNJ; ++j) {
          if constexpr (IS_INT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const i32x4 av = {(int)af[i][h].x, (int)af[i][h].y, (int)af[i][h].z, (int)af[i][h].w};
              const i32x4 bv = {(int)bf[j][h].x, (int)bf[j][h].y, (int)bf[j][h].z, (int)bf[j][h].w};
              acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[i][j], 0, 0, 0);
            }
          } else {
            const i32x8 av = {(int)af[i][0].x, (int)af[i][0].y, (int)af[i][0].z, (int)af[i][0].w,
                              (int)af[i][1].x, (int)af[i][1].y, (int)af[i][1].z, (int)af[i][1].w};
            const i32x8 bv = {(int)bf[j][0].x, (int)bf[j][0].y, (int)bf[j][0].z, (int)bf[j][0].w,
                              (int)bf[j][1].x, (int)bf[j][1].y, (int)bf[j][1].z, (int)bf[j][1].w};
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 0, 0, 0, 127, 0, 127);
          }
        }
    }
    // next tile landed (this wave's DMAs) and everybody is done reading `buf`
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // epilogue: C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNJ; ++j) {
      const int gn = n0 + wn * (TNJ * 32) + j * 32 + (lane & 31);
      if (gn >= p.N) continue;
      float cs = 1.f, bias = 0.f;
      if (EPI != EPI_INT32 && EPI != EPI_FP8_RAW) {
        cs = p.col_scale[gn];
        if (p.bias != nullptr) bias = bf16_lo_to_f32(p.bias[gn]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm >= p.M) continue;
        if (EPI == EPI_INT32) {
          reinterpret_cast<int32_t*>(p.out)[(size_t)gm * p.N + gn] = (int32_t)acc[i][j][r];
        } else if (EPI == EPI_FP8_RAW) {
          reinterpret_cast<float*>(p.out)[(size_t)gm * p.N + gn] = (float)acc[i][j][r];
        } else if (EPI == EPI_INT8_SCALED) {
          const float t = round_bf16((float)acc[i][j][r] * p.row_scale[gm]);
          float y = t * cs;
          if (p.bias != nullptr) y += bias;
          reinterpret_cast<uint16_t*>(p.out)[(size_t)gm * p.N + gn] = f32_to_bf16_bits(y);
        } else {
          float y = (float)acc[i][j][r] * p.row_scale[gm] * cs;
          if (p.bias != nullptr) y += bias;
          reinterpret_cast<uint16_t*>(p.out)[(size_t)gm * p.N + gn] = f32_to_bf16_bits(y);
        }
      }
    }
}

thread_local bool g_gemm8_force_regstage = false;  // profiling: ao_gemm8_set_variant(1)
thread_local bool g_gemm8_tiled_only = false;      // profiling / A-B tests: ao_gemm8_set_variant(100) -- never a weight-streaming kernel

thread_local int g_gemm8_tm = 0;  // profiling: 0 = by shape, 2 / 4 = force

template <int EPI, int TM, int WN, int TNJ = 2>
int launch_gemm8_dma_tm(const Gemm8Args& p, hipStream_t stream) {
  constexpr int WGM = TM * 64, WGN = WN * TNJ * 32;
  dim3 grid((unsigned)((p.N + WGN - 1) / WGN), (unsigned)((p.M + WGM - 1) / WGM)), block(128 * WN);
  const size_t smem = 2 * (size_t)(WGM + WGN) * BK;  // 64 KiB (128 x 128) ... 128 KiB (256 x 256)
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_dma_kernel<EPI, TM, WN, TNJ>), smem, "hipFuncSetAttribute(gemm8_dma_kernel)")) return rc;
  ao::launch(gemm8_dma_kernel<EPI, TM, WN, TNJ>, grid, block, smem, stream, p);
  AO_LAUNCH_CHECK("gemm8_dma_kernel launch");
  return AO_OK;
}

// the shapes the phase-interleaved 256 x 256 kernel takes from the two-stage tile kernels (product rule; see launch_gemm8_dma)
bool gemm8_p8_band(int64_t M, int64_t N, int64_t K) {
  const int64_t big = ((N + 255) / 256) * ((M + 255) / 256), t128 = ((N + 127) / 128) * ((M + 127) / 128);
  return big >= 160 || (big >= 128 && (K <= 4096 || t128 > 512));
}

template <int EPI>
int launch_gemm8_dma(const Gemm8Args& p, hipStream_t stream) {
  // variants (ao_gemm8_set_variant): 2 = 128 x 128 tile / 4 waves; 4 = 256 x 128 / 4 waves (one wave per SIMD:
  // measured 0.75-0.95x, nothing hides its ds_read -> MFMA latency); 8 = 256 x 256 / 8 waves.
  if (g_gemm8_tm == 4) return launch_gemm8_dma_tm<EPI, 4, 2>(p, stream);
  if (g_gemm8_tm == 16) return launch_gemm8_dma_tm<EPI, 4, 2, 4>(p, stream);  // 256 x 256, 4 waves of 128 x 128
  // (A 4-stage, 64-byte-K-step pipeline with hand-counted vmcnt was measured at 0.94-0.97x of these two-stage
  // kernels at both tile shapes, profiles/bench_8bit_r01_gemm.txt: the loop is LDS-read bound, not latency bound.)
  const int64_t big = (int64_t)((p.N + 255) / 256) * ((p.M + 255) / 256);
  // The phase-interleaved 256 x 256 kernel (gemm8_p8_kernels.hip; variant 32 forces it) once the problem has >= 160 such tiles.
  // Measured on the Llama-3-8B shapes (profiles/gemm8_variants_r02.txt, TOP/s int8, this kernel vs the two-stage kernels):
  // M = 8192: 2046 / 2084 / 2492 / 2796 vs 1557 / 1508 / 1777 / 2153; M = 2048: qkv (192 tiles) 1832 vs 1210, gate_up 2148 vs 1566,
  // but o / down (128 tiles: half the CUs idle) 1350 / 1710 vs 1330 / 1834; M = 512: gate_up (224 tiles) 2122 vs 1231.
  // Round 4 (profiles/gemm8_p8_band_r04.jsonl, cold weights): at 128 .. 159 such tiles it also wins while K <= 4096 (1024 x 8192 x 1024: 24.3 -> 20.2 us;
  // K = 8192 / 14336: 2 - 7 % behind) and whenever the 128 x 128 grid would need a second round of the chip (> 512 tiles: 1280 x 7168 x 8192 129 -> 84 us)
  if ((g_gemm8_tm == 33 && gemm8_p8_fits(p.M, p.N, p.K)) || (g_gemm8_tm == 0 && gemm8_p8h_band(p.M, p.N, p.K)))  // the 256 x 128 phase-interleaved form (33 forces it)
    return gemm8_p8h((int)EPI, p.a, p.b, p.row_scale, p.col_scale, p.bias, p.out, p.M, p.N, p.K, stream);
  if ((g_gemm8_tm == 32 || (g_gemm8_tm == 0 && gemm8_p8_band(p.M, p.N, p.K))) && gemm8_p8_fits(p.M, p.N, p.K))
    return gemm8_p8((int)EPI, p.a, p.b, p.row_scale, p.col_scale, p.bias, p.out, p.M, p.N, p.K, stream);
  if (g_gemm8_tm == 8 || (g_gemm8_tm == 0 && big >= 512)) return launch_gemm8_dma_tm<EPI, 4, 4>(p, stream);
  return launch_gemm8_dma_tm<EPI, 2, 2>(p, stream);
}

template <int EPI>
int launch_gemm8(const Gemm8Args& p, hipStream_t stream) {
  if (p.K % BK == 0 && !g_gemm8_force_regstage) return launch_gemm8_dma<EPI>(p, stream);  // K % 128 == 0 (the pipelined kernels need K % 64)
  dim3 grid((unsigned)((p.N + BN - 1) / BN), (unsigned)((p.M + BM - 1) / BM)), block(THREADS);
  const size_t smem = 2 * 2 * TILE_BYTES;  // 73,728 B
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm8_kernel<EPI>), smem, "hipFuncSetAttribute(gemm8_kernel)")) return rc;
  ao::launch(gemm8_kernel<EPI>, grid, block, smem, stream, p);
  AO_LAUNCH_CHECK("gemm8_kernel launch");
  return AO_OK;
}

int check_gemm_shape(const char* fn, int64_t M, int64_t N, int64_t K) {
  AO_REQUIRE(M >= 0 && N > 0 && K > 0, "%s: bad shape M=%lld N=%lld K=%lld", fn, (long long)M, (long long)N, (long long)K);
  AO_REQUIRE(K % 16 == 0, "%s: K=%lld must be a multiple of 16", fn, (long long)K);
  AO_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "%s: dimension too large", fn);
  AO_REQUIRE((M + BM - 1) / BM <= 65535, "%s: M=%lld too large for one launch", fn, (long long)M);
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int ao_gemm8_set_variant(int variant) {
  g_gemm8_force_regstage = (variant == 1);
  g_gemm8_tiled_only = (variant == 100);
  g_mx_variant = (variant == 110) ? 1 : (variant == 111) ? 2 : 0;
  // MX decode groups: 113 one workgroup per tile (the form of larger groups); 129 the stream-K kernel's per-step-scales form (8 waves, two per CU:
  // what K % 512 != 0 takes) on every K
  mx_rb_set_stream(variant == 113 ? 0 : 1, variant != 129);
  // the straight-line decode kernel (dec8_kernels.hip): 201 / 202 / 204 / 207 / 208 force its ring depth, 290 half-line loads, 299 never
  g_dec8_mode = (variant >= 200 && variant <= 299) ? variant : 0;
  // the register-ring mid-M kernel (mid8_kernels.hip): 300 never, 301 wherever the shape allows, 310 + S: S K-parts forced
  g_mid8_mode = (variant >= 300 && variant <= 329) ? variant : 0;
  g_gemm8_tm = (variant == 2 || variant == 4 || variant == 8 || variant == 16 || variant == 32 || variant == 33) ? variant : 0;
  // the fp8 weight-streaming mid-M kernel: 101 always, 100 or any explicit GEMM variant never, 0 by shape
  fp8_rowwise_rb_set_mode(variant == 101 ? 2 : variant == 102 ? 3 : (variant != 0 && variant < 110 && variant != 103) ? 1 : 0);
  rb8_set_wave_grid(variant != 103);  // 103: the weight-streaming kernel's round-3 wave arrangement (1 x 8), product dispatch otherwise
  return AO_OK;
}

namespace ao { namespace { thread_local int g_tune[16] = {0}; } }
extern "C" int ao_gemm8_set_tuning(int key, int value) {
  AO_REQUIRE(key >= 1 && key <= 9, "ao_gemm8_set_tuning: unknown key %d", key);
  g_tune[key] = value;
  mx_stream_set_tuning(g_tune[9]);
  rb8_set_tuning(g_tune[1], g_tune[2], g_tune[5]);
  rb8_set_slab_rows(g_tune[3]);
  gemm8_p8_set_group_rows(g_tune[4]);
  gemm8_p8_set_split(g_tune[7]);
  gemm8_p8_set_persistent(g_tune[6]);
  gemm8_p8h_set_form(g_tune[8]);
  return AO_OK;
}

extern "C" int ao_int8_scaled_mm(const int8_t* xq, const float* x_scale, const int8_t* wq, const float* w_scale,
                                 const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K, void* stream) {
  if (int rc = check_gemm_shape(__func__, M, N, K)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(xq);
  AO_REQUIRE_PTR(x_scale);
  AO_REQUIRE_PTR(wq);
  AO_REQUIRE_PTR(w_scale);
  AO_REQUIRE_PTR(y);
  // weight-bandwidth-bound sizes stream the weights once, as for fp8 below: M <= 32 per-tile streaming kernel
  // (stream8_kernels.hip), few output tiles the LDS-staged one (rb8_kernels.hip); Llama-3-8B down_proj at M = 1 took 77 us
  // through the tiled GEMM
  if (M <= 16 && !fp8_rowwise_rb_forced() && !g_gemm8_tiled_only && dec8_takes(M, N, K))
    return dec8_scaled(true, xq, x_scale, wq, w_scale, bias, y, M, N, K, (hipStream_t)stream);  // round 4: full-line register ring
  if (M > 16 && !fp8_rowwise_rb_forced() && !g_gemm8_tiled_only && g_gemm8_tm == 0 && !g_gemm8_force_regstage && mid8_takes(M, N, K))
    return mid8_scaled(true, xq, x_scale, wq, w_scale, bias, y, M, N, K, (hipStream_t)stream);  // round 4: 16 < M <= 256, few output tiles
  const bool rb_small = !g_gemm8_tiled_only && g_gemm8_tm == 0 && !g_gemm8_force_regstage && rb8_small_m_preferred(M, N, K);  // round 6
  if (N % 16 == 0 && K % 128 == 0 && M <= 32 && !rb_small && !fp8_rowwise_rb_forced() && !g_gemm8_tiled_only)
    return int8_scaled_stream(xq, wq, x_scale, w_scale, bias, y, M, N, K, (hipStream_t)stream);
  if (N % 16 == 0 && (rb_small || fp8_rowwise_rb_preferred(M, N, K))) return int8_scaled_rb(xq, wq, x_scale, w_scale, bias, y, M, N, K, (hipStream_t)stream);
  Gemm8Args p{reinterpret_cast<const uint8_t*>(xq), reinterpret_cast<const uint8_t*>(wq), x_scale, w_scale, bias, y,
              (int)M, (int)N, (int)K};
  return launch_gemm8<EPI_INT8_SCALED>(p, (hipStream_t)stream);
}

// Which kernel the product dispatch of ao_fp8_scaled_mm / ao_int8_scaled_mm takes for a shape (host logic only: no launch, no GPU) -- the
// rules of the two entry points above and below, restated without the tuning overrides.  tests/test_host_dispatch.py pins the table.
extern "C" const char* ao_gemm8_kernel_name(int int8, int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 16 != 0) return "invalid";
  if (M <= 16 && dec8_takes(M, N, K)) return "dec8_kernel";
  if (M > 16 && mid8_takes(M, N, K)) return "mid8_kernel";
  const bool rb_small = rb8_small_m_preferred(M, N, K);
  const bool rb_shape = N % 16 == 0 && (rb_small || fp8_rowwise_rb_preferred(M, N, K));
  if (int8) {
    if (N % 16 == 0 && K % 128 == 0 && M <= 32 && !rb_small) return "stream8_kernel";
    if (rb_shape) return "rb8_kernel";
  } else {
    if (N % 16 != 0) return "invalid";
    const bool rb = rb_shape && (M > 32 || rb_small);
    if (K % 128 == 0 && ((M <= 32 && !rb_small) || (M <= 64 && !rb))) return "stream8_kernel";
    if (rb) return "rb8_kernel";
  }
  if (K % BK != 0) return "gemm8_kernel";  // register-staged tiles (K % 128 != 0)
  const int64_t big = ((N + 255) / 256) * ((M + 255) / 256);
  if (gemm8_p8h_band(M, N, K)) return "gemm8_p8h_kernel";
  if (gemm8_p8_band(M, N, K) && gemm8_p8_fits(M, N, K)) return gemm8_p8_persistent_shape(M, N, K) ? "gemm8_p8p_kernel" : "gemm8_p8_kernel";
  return big >= 512 ? "gemm8_dma_kernel<256x256>" : "gemm8_dma_kernel<128x128>";
}

// The launch shape behind ao_gemm8_kernel_name: tile rows, column-tile width and K parts of the product dispatch (host logic only).
// rb8_kernel: the cost model's pick (64- / 128-row slabs, 32 / 64 / 128 columns, 1 .. 8 parts); gemm8_p8h_kernel: 256 x 128, 1 .. 4 parts;
// every other kernel: its tile, one part.
static int gemm8_plan3(int int8, int64_t M, int64_t N, int64_t K, int* tile_rows, int* tile_cols, int* k_parts) {
  const std::string name = ao_gemm8_kernel_name(int8, M, N, K);
  AO_REQUIRE(name != "invalid", "ao_gemm8_plan: no kernel takes M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  *k_parts = 1;
  if (name == "rb8_kernel") rb8_plan_query(M, N, K, tile_rows, tile_cols, k_parts);
  else if (name == "gemm8_p8h_kernel") { *tile_rows = 256; *tile_cols = 128; *k_parts = gemm8_p8h_parts(M, N, K); }
  else if (name == "gemm8_p8_kernel" || name == "gemm8_p8p_kernel" || name == "gemm8_dma_kernel<256x256>") *tile_rows = *tile_cols = 256;
  else if (name == "gemm8_dma_kernel<128x128>" || name == "gemm8_kernel") *tile_rows = *tile_cols = 128;
  else { *tile_rows = 16; *tile_cols = 16; }  // the per-tile streaming kernels (dec8 / mid8 / stream8): 16-wide n-tiles, K split among the waves of a workgroup
  return AO_OK;
}
extern "C" int ao_gemm8_plan(int int8, int64_t M, int64_t N, int64_t K, int* tile_cols, int* k_parts) {
  AO_REQUIRE_PTR(tile_cols);
  AO_REQUIRE_PTR(k_parts);
  int rows = 0;
  return gemm8_plan3(int8, M, N, K, &rows, tile_cols, k_parts);
}
// (round 6) the tile's rows as well: the weight-streaming kernel's slab height is part of the plan since 64-row slabs serve M > 64
extern "C" int ao_gemm8_plan_rows(int int8, int64_t M, int64_t N, int64_t K, int* tile_rows) {
  AO_REQUIRE_PTR(tile_rows);
  int cols = 0, parts = 0;
  return gemm8_plan3(int8, M, N, K, tile_rows, &cols, &parts);
}

extern "C" int ao_int8_int_mm(const int8_t* a, const int8_t* b_t, int32_t* c, int64_t M, int64_t N, int64_t K,
                              void* stream) {
  if (int rc = check_gemm_shape(__func__, M, N, K)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(a);
  AO_REQUIRE_PTR(b_t);
  AO_REQUIRE_PTR(c);
  Gemm8Args p{reinterpret_cast<const uint8_t*>(a), reinterpret_cast<const uint8_t*>(b_t), nullptr, nullptr, nullptr, c,
              (int)M, (int)N, (int)K};
  return launch_gemm8<EPI_INT32>(p, (hipStream_t)stream);
}

// Unscaled e4m3 x e4m3 products, fp32 out: the partial sums a K-sharded (row-parallel) fp8 linear all-reduces before
// ao_fp8_scale_epilogue applies scale_a[m] * scale_b[n] (+ bias) once, like the unsharded aten::_scaled_mm.
extern "C" int ao_fp8_mm_f32(const uint8_t* a, const uint8_t* b, float* c, int64_t M, int64_t N, int64_t K, void* stream) {
  if (int rc = check_gemm_shape(__func__, M, N, K)) return rc;
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(a);
  AO_REQUIRE_PTR(b);
  AO_REQUIRE_PTR(c);
  Gemm8Args p{a, b, nullptr, nullptr, nullptr, c, (int)M, (int)N, (int)K};
  return launch_gemm8<EPI_FP8_RAW>(p, (hipStream_t)stream);
}

extern "C" int ao_fp8_scaled_mm(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b,
                                const uint16_t* bias, uint16_t* y, int64_t M, int64_t N, int64_t K, void* stream) {
  if (int rc = check_gemm_shape(__func__, M, N, K)) return rc;
  AO_REQUIRE(N % 16 == 0, "ao_fp8_scaled_mm: N=%lld must be a multiple of 16", (long long)N);
  if (M == 0) return AO_OK;
  AO_REQUIRE_PTR(a);
  AO_REQUIRE_PTR(b);
  AO_REQUIRE_PTR(scale_a);
  AO_REQUIRE_PTR(scale_b);
  AO_REQUIRE_PTR(y);
  // Weight-bandwidth-bound sizes stream the weights once instead of tiling a GEMM (Llama-70B TP8 shards, us per call,
  // profiles/bench_8bit_r01_fp8.jsonl):
  //   * M <= 32: stream8_kernels.hip -- activations straight from L2 per wave, no LDS staging, no split-K
  //     (gate_up 7168x8192: 15 us at M = 1, 20 at 16; the LDS-staged kernel below needs 23);
  //   * 32 < M and too few 128 x 128 tiles to fill the chip: rb8_kernels.hip (24 us at M = 64 where the kernel above
  //     needs 40; 27 us at M = 128 where the tiled GEMM needs 55);
  //   * otherwise the tiled LDS-DMA GEMM.
  // (at 32 < M <= 64 it only wins once K is long enough to amortise its start-up and split-K meeting: o_proj shard 8192x1024 8.5 vs 10.9 us)
  if (M <= 16 && !fp8_rowwise_rb_forced() && !g_gemm8_tiled_only && dec8_takes(M, N, K))
    return dec8_scaled(false, a, scale_a, b, scale_b, bias, y, M, N, K, (hipStream_t)stream);  // round 4: full-line register ring
  if (M > 16 && !fp8_rowwise_rb_forced() && !g_gemm8_tiled_only && g_gemm8_tm == 0 && !g_gemm8_force_regstage && mid8_takes(M, N, K))
    return mid8_scaled(false, a, scale_a, b, scale_b, bias, y, M, N, K, (hipStream_t)stream);  // round 4: 16 < M <= 256, few output tiles
  // (round 4, cold weights -- every call of the replay reads another copy: the LDS-staged kernel wins from 33 rows on at every K of the
  // 70B / TP8 shards, down 8192 x 3584 at M = 64: 14.0 us against 20.9 through the per-tile kernel, o 8192 x 1024: 8.5 against 9.2; the
  // round-1 rule -- "only from K >= 4096 at 32 < M <= 64" -- had been measured on ONE re-used weight, i.e. out of the Infinity Cache)
  const bool rb_small = !g_gemm8_tiled_only && g_gemm8_tm == 0 && !g_gemm8_force_regstage && rb8_small_m_preferred(M, N, K);  // round 6
  const bool rb = rb_small || (fp8_rowwise_rb_preferred(M, N, K) && (M > 32 || fp8_rowwise_rb_forced()));
  if (K % 128 == 0 && ((M <= 32 && !rb_small) || (M <= 64 && !rb)) && !(fp8_rowwise_rb_forced() && rb) && !g_gemm8_tiled_only)
    return fp8_rowwise_stream(a, b, scale_a, scale_b, bias, y, M, N, K, (hipStream_t)stream);
  if (rb) return fp8_rowwise_rb(a, b, scale_a, scale_b, bias, y, M, N, K, (hipStream_t)stream);
  Gemm8Args p{a, b, scale_a, scale_b, bias, y, (int)M, (int)N, (int)K};
  return launch_gemm8<EPI_FP8_ROWWISE>(p, (hipStream_t)stream);
}
