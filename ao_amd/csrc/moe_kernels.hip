// MoE token-group padding for gfx950: the glue either side of the MXFP8 grouped GEMM.
//
// Replaces torchao::fused_pad_token_groups / torchao::fused_unpad_token_groups (schemas
// torchao/prototype/moe_training/kernels/mxfp8/quant.py:1244-1246, 1319-1321), with the semantics of the
// reference's own torch restatement torch_pad_token_groups / torch_unpad_token_groups (quant.py:368-480):
//   pad:   group g (rows offs[g-1] .. offs[g]) is copied to rows pstart[g] .. of a zero-filled output whose
//          groups start at multiples of `alignment`; output rows = align_up(num_tokens + E * alignment);
//          also returns pstart[E] and pend[E] (int32, cumulative padded sizes).
//   unpad: the inverse gather.
// Pure HBM-bound row copies: 2 B moved per payload byte (+ zero fill).  One wave per row, 16-byte lanes,
// four rows' worth of loads in flight per wave; the group of a row is found with one ballot over the
// (L2-resident) offsets, 64 groups per pass.
#include "common.h"

namespace ao {
namespace {

constexpr int kRowsPerWg = 4;  // one wave per row

// padded group bounds: one wave, 64 groups per pass, running prefix carried across passes
__global__ __launch_bounds__(64) void moe_padded_offsets_kernel(const int32_t* __restrict__ offs, int32_t* __restrict__ pstart,
                                                                int32_t* __restrict__ pend, int E, int alignment) {
  const int lane = threadIdx.x;
  int carry = 0;
  for (int g0 = 0; g0 < E; g0 += 64) {
    const int g = g0 + lane;
    int padded = 0;
    if (g < E) {
      const int size = offs[g] - (g > 0 ? offs[g - 1] : 0);
      padded = ((size + alignment - 1) / alignment) * alignment;
    }
    int incl = padded;  // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (g < E) {
      pend[g] = carry + incl;
      pstart[g] = carry + incl - padded;
    }
    carry += __shfl(incl, 63);
  }
}

// first group g with row < ends[g], or E
__device__ __forceinline__ int find_group(const int32_t* __restrict__ ends, int E, int row, int lane) {
  for (int g0 = 0; g0 < E; g0 += 64) {
    const int g = g0 + lane;
    const bool hit = g < E && row < ends[g];
    const unsigned long long m = __ballot(hit);
    if (m) return g0 + __builtin_ctzll(m);
  }
  return E;
}

template <typename V>
__device__ __forceinline__ void copy_row(const V* __restrict__ src, V* __restrict__ dst, int64_t nvec, int lane) {
  int64_t i = lane;
  for (; i + 192 < nvec; i += 256) {
    const V a = src[i], b = src[i + 64], c = src[i + 128], d = src[i + 192];
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + 64);
    __builtin_nontemporal_store(c, dst + i + 128);
    __builtin_nontemporal_store(d, dst + i + 192);
  }
  for (; i < nvec; i += 64) __builtin_nontemporal_store(src[i], dst + i);
}

template <typename V>
__device__ __forceinline__ void zero_row(V* __restrict__ dst, int64_t nvec, int lane) {
  V z;
  __builtin_memset(&z, 0, sizeof(V));
  for (int64_t i = lane; i < nvec; i += 64) __builtin_nontemporal_store(z, dst + i);
}

// V = u32x4 / uint32_t / uint16_t: the widest unit that divides the row
template <typename V>
__global__ __launch_bounds__(64 * kRowsPerWg) void moe_pad_rows_kernel(const V* __restrict__ in, const int32_t* __restrict__ offs,
                                                                       const int32_t* __restrict__ pstart,
                                                                       const int32_t* __restrict__ pend, V* __restrict__ out,
                                                                       int64_t out_rows, int64_t nvec, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerWg + (threadIdx.x >> 6);
  if (row >= out_rows) return;
  V* dst = out + row * nvec;
  const int g = (row < 0x7fffffff) ? find_group(pend, E, (int)row, lane) : E;
  if (g < E) {
    const int first = g > 0 ? offs[g - 1] : 0;
    const int local = (int)row - pstart[g];
    if (local < offs[g] - first) {
      copy_row(in + (int64_t)(first + local) * nvec, dst, nvec, lane);
      return;
    }
  }
  zero_row(dst, nvec, lane);  // alignment padding, or past the last group
}

template <typename V>
__global__ __launch_bounds__(64 * kRowsPerWg) void moe_unpad_rows_kernel(const V* __restrict__ in, const int32_t* __restrict__ offs,
                                                                         const int32_t* __restrict__ pstart, V* __restrict__ out,
                                                                         int64_t num_tokens, int64_t nvec, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerWg + (threadIdx.x >> 6);
  if (row >= num_tokens) return;
  V* dst = out + row * nvec;
  const int g = find_group(offs, E, (int)row, lane);
  if (g < E) {
    const int first = g > 0 ? offs[g - 1] : 0;
    copy_row(in + (int64_t)(pstart[g] + ((int)row - first)) * nvec, dst, nvec, lane);
  } else {
    zero_row(dst, nvec, lane);  // num_tokens > offs[E-1]: the reference raises after a host sync; no rows to take
  }
}

int check_common(const char* fn, int64_t rows, int64_t dim, int elem_bytes, int64_t E, int alignment) {
  AO_REQUIRE(rows >= 0 && dim > 0, "%s: bad shape rows=%lld dim=%lld", fn, (long long)rows, (long long)dim);
  AO_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "%s: inputs must be bfloat16 or float32 (element size 2 or 4), got %d", fn, elem_bytes);
  AO_REQUIRE(E > 0 && E < (1 << 20), "%s: number of groups %lld out of range", fn, (long long)E);
  AO_REQUIRE(alignment > 0, "%s: alignment_size must be positive, got %d", fn, alignment);
  AO_REQUIRE(rows < (1ll << 31) - (E + 1) * (int64_t)alignment, "%s: row count must fit int32", fn);
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int64_t ao_moe_padded_rows(int64_t num_tokens, int64_t num_groups, int alignment) {
  if (alignment <= 0) return -1;
  const int64_t rows = num_tokens + num_groups * alignment;
  return ((rows + alignment - 1) / alignment) * alignment;
}

extern "C" int ao_moe_pad_token_groups(const void* inputs, const int32_t* offsets, void* padded, int32_t* padded_starts,
                                       int32_t* padded_ends, int64_t num_tokens, int64_t dim, int elem_bytes, int64_t num_groups,
                                       int alignment, void* stream) {
  AO_REQUIRE_PTR(offsets);
  AO_REQUIRE_PTR(padded);
  AO_REQUIRE_PTR(padded_starts);
  AO_REQUIRE_PTR(padded_ends);
  if (int rc = check_common(__func__, num_tokens, dim, elem_bytes, num_groups, alignment)) return rc;
  if (num_tokens > 0) AO_REQUIRE_PTR(inputs);
  hipStream_t st = static_cast<hipStream_t>(stream);
  ao::launch(moe_padded_offsets_kernel, dim3(1), dim3(64), 0, st, offsets, padded_starts, padded_ends, (int)num_groups, alignment);
  AO_LAUNCH_CHECK("moe_padded_offsets_kernel launch");
  const int64_t out_rows = ao_moe_padded_rows(num_tokens, num_groups, alignment);
  const int64_t row_bytes = dim * elem_bytes;
  const dim3 grid((unsigned)((out_rows + kRowsPerWg - 1) / kRowsPerWg)), block(64 * kRowsPerWg);
  const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)inputs % 16 == 0) && ((uintptr_t)padded % 16 == 0);
  if (a16)
    ao::launch(moe_pad_rows_kernel<u32x4>, grid, block, 0, st, static_cast<const u32x4*>(inputs), offsets, padded_starts, padded_ends,
               static_cast<u32x4*>(padded), out_rows, row_bytes / 16, (int)num_groups);
  else if (row_bytes % 4 == 0)
    ao::launch(moe_pad_rows_kernel<uint32_t>, grid, block, 0, st, static_cast<const uint32_t*>(inputs), offsets, padded_starts, padded_ends,
               static_cast<uint32_t*>(padded), out_rows, row_bytes / 4, (int)num_groups);
  else
    ao::launch(moe_pad_rows_kernel<uint16_t>, grid, block, 0, st, static_cast<const uint16_t*>(inputs), offsets, padded_starts, padded_ends,
               static_cast<uint16_t*>(padded), out_rows, row_bytes / 2, (int)num_groups);
  AO_LAUNCH_CHECK("moe_pad_rows_kernel launch");
  return AO_OK;
}

extern "C" int ao_moe_unpad_token_groups(const void* padded, const int32_t* offsets, const int32_t* padded_starts, void* out,
                                         int64_t num_tokens, int64_t dim, int elem_bytes, int64_t num_groups, void* stream) {
  AO_REQUIRE_PTR(offsets);
  AO_REQUIRE_PTR(padded_starts);
  if (int rc = check_common(__func__, num_tokens, dim, elem_bytes, num_groups, 1)) return rc;
  if (num_tokens == 0) return AO_OK;
  AO_REQUIRE_PTR(padded);
  AO_REQUIRE_PTR(out);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t row_bytes = dim * elem_bytes;
  const dim3 grid((unsigned)((num_tokens + kRowsPerWg - 1) / kRowsPerWg)), block(64 * kRowsPerWg);
  const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)padded % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (a16)
    ao::launch(moe_unpad_rows_kernel<u32x4>, grid, block, 0, st, static_cast<const u32x4*>(padded), offsets, padded_starts,
               static_cast<u32x4*>(out), num_tokens, row_bytes / 16, (int)num_groups);
  else if (row_bytes % 4 == 0)
    ao::launch(moe_unpad_rows_kernel<uint32_t>, grid, block, 0, st, static_cast<const uint32_t*>(padded), offsets, padded_starts,
               static_cast<uint32_t*>(out), num_tokens, row_bytes / 4, (int)num_groups);
  else
    ao::launch(moe_unpad_rows_kernel<uint16_t>, grid, block, 0, st, static_cast<const uint16_t*>(padded), offsets, padded_starts,
               static_cast<uint16_t*>(out), num_tokens, row_bytes / 2, (int)num_groups);
  AO_LAUNCH_CHECK("moe_unpad_rows_kernel launch");
  return AO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// E8M0 block scales -> the "128 x 4 blocked" layout, per token group (round 6; SURVEY.md 8 f4 "scale re-layout").
//
// Replaces torchao::mx_block_rearrange_2d_M_groups (schema torchao/prototype/moe_training/kernels/mxfp8/quant.py:969-973, host wrapper
// csrc/cuda/mx_kernels/mxfp8_extension.cpp:178-300, kernels mx_block_rearrange_2d_M_groups.cu:165-470) with the semantics of the
// reference's own torch restatement torch_to_blocked_2d_M_groups (quant.py:136-196) over to_blocked (prototype/mx_formats/utils.py:31-72):
//   in   [rows][cols] bytes, row groups ending at offs[g];
//   out  [rows + 128 G][pcols], pcols = 4 ceil(cols / 4), zero wherever nothing below lands;
//   group g starts at out row start_g = sum over h < g of 128 ceil(size_h / 128) and is written as 512-byte tiles: tile (rb, cb) -- rows
//   128 rb .. + 127 of the group, columns 4 cb .. + 3 -- at byte offset start_g pcols + (rb ncb + cb) 512, element (r, c) of the tile at
//   (r % 32) 16 + (r / 32) 4 + c; rows past the group and columns past `cols` read as zero.
// The MI355X GEMMs of this library take row-major scales (the scaled MFMA reads them from VGPRs); this op exists for callers that hold
// the reference's blocked layout as a data format (checkpoints, an exchange with an NVIDIA peer).  HBM-bound byte shuffle, 2 B moved per
// scale byte: a workgroup owns one 128-row block x 64 columns, reads it as row-contiguous dwords into LDS and writes its 16 tiles as
// 16-byte pieces (thread (tile, q): rows q, q + 32, q + 64, q + 96 of column block `tile` are one piece).
namespace {

constexpr int kBlkCols = 64;  // columns per workgroup (16 column blocks)

template <bool DW>
__global__ __launch_bounds__(256) void mx_blocked_m_groups_kernel(const uint8_t* __restrict__ in, const int32_t* __restrict__ offs,
                                                                  uint8_t* __restrict__ out, int rows, int cols, int G, long long out_bytes,
                                                                  int pcols) {
  __shared__ uint32_t tile[128][17];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, cb0 = blockIdx.x * (kBlkCols / 4), ncb = pcols >> 2;
  // the group of padded row block b (uniform: scalar loads; offs == nullptr: one group of all rows)
  int begin = 0, end = 0, rb = 0;
  {
    int prev = 0, blocks = 0;
    bool found = false;
    for (int g = 0; g < G && !found; ++g) {
      const int e = offs != nullptr ? min(max(offs[g], prev), rows) : rows;
      const int nb = (e - prev + 127) >> 7;
      if (b < blocks + nb) { found = true; begin = prev; end = e; rb = b - blocks; }
      blocks += nb;
      prev = e;
    }
  }
  const int row0 = begin + rb * 128;  // (no group: begin = end = 0 -> every row reads as zero)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 256 + tid, r = idx >> 4, c = idx & 15;
    const int src = row0 + r, col = (cb0 + c) * 4;
    uint32_t v = 0u;
    if (src < end && col < cols) {
      const uint8_t* p = in + (size_t)src * cols + col;
      if (DW) {
        v = *reinterpret_cast<const uint32_t*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j < cols) v |= (uint32_t)p[j] << (8 * j);
      }
    }
    tile[r][c] = v;
  }
  __syncthreads();
  const long long base = (long long)b * 128 * pcols;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = i * 256 + tid, t = idx >> 5, q = idx & 31;
    const long long off = base + (long long)(cb0 + t) * 512 + q * 16;
    if (cb0 + t < ncb && off < out_bytes) {
      const u32x4 v = {tile[q][t], tile[q + 32][t], tile[q + 64][t], tile[q + 96][t]};
      if (off + 16 <= out_bytes) {
        *reinterpret_cast<u32x4*>(out + off) = v;
      } else {  // the upper-bound buffer ends inside this piece (its row count is not a multiple of 128): the bytes it still has
        for (int j = 0; off + j < out_bytes; ++j) out[off + j] = (uint8_t)(v[j >> 2] >> (8 * (j & 3)));
      }
    }
  }
}

int mx_blocked_launch(const char* fn, const uint8_t* scales, const int32_t* offsets, uint8_t* out, int64_t rows, int64_t cols, int64_t G,
                      int64_t out_rows, hipStream_t st) {
  AO_REQUIRE(rows >= 0 && cols > 0 && rows < (1ll << 31) - 128 * (G + 1) && cols < (1ll << 24), "%s: rows=%lld cols=%lld out of range", fn,
             (long long)rows, (long long)cols);
  const int64_t pcols = (cols + 3) / 4 * 4;
  if (out_rows == 0) return AO_OK;
  AO_REQUIRE_PTR(out);
  AO_REQUIRE(rows == 0 || scales != nullptr, "%s: null scales", fn);
  AO_REQUIRE((uintptr_t)out % 16 == 0, "%s: the output must be 16-byte aligned", fn);
  const int64_t blocks = (out_rows + 127) / 128;
  AO_REQUIRE(blocks <= 65535, "%s: %lld row blocks exceed one launch", fn, (long long)blocks);
  const dim3 grid((unsigned)((pcols + kBlkCols - 1) / kBlkCols), (unsigned)blocks), block(256);
  const bool dw = cols % 4 == 0 && (uintptr_t)scales % 4 == 0;
  if (dw)
    ao::launch(mx_blocked_m_groups_kernel<true>, grid, block, 0, st, scales, offsets, out, (int)rows, (int)cols, (int)G,
               (long long)(out_rows * pcols), (int)pcols);
  else
    ao::launch(mx_blocked_m_groups_kernel<false>, grid, block, 0, st, scales, offsets, out, (int)rows, (int)cols, (int)G,
               (long long)(out_rows * pcols), (int)pcols);
  AO_LAUNCH_CHECK("mx_blocked_m_groups_kernel launch");
  return AO_OK;
}

}  // namespace

extern "C" int64_t ao_mx_blocked_rows(int64_t rows, int64_t num_groups) { return rows + 128 * num_groups; }

extern "C" int ao_mx_block_rearrange_2d_m_groups(const uint8_t* scales, const int32_t* offsets, uint8_t* out, int64_t rows, int64_t cols,
                                                 int64_t num_groups, void* stream) {
  AO_REQUIRE_PTR(offsets);
  AO_REQUIRE(num_groups > 0 && num_groups < (1 << 16), "%s: number of groups %lld out of range", __func__, (long long)num_groups);
  return mx_blocked_launch(__func__, scales, offsets, out, rows, cols, num_groups, ao_mx_blocked_rows(rows, num_groups),
                           static_cast<hipStream_t>(stream));
}

extern "C" int ao_mx_to_blocked(const uint8_t* scales, uint8_t* out, int64_t rows, int64_t cols, void* stream) {
  return mx_blocked_launch(__func__, scales, nullptr, out, rows, cols, 1, (rows + 127) / 128 * 128, static_cast<hipStream_t>(stream));
}
