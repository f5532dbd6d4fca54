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
