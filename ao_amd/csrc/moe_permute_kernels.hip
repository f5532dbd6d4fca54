// Expert-parallel token regrouping for gfx950: rank-major -> expert-major order with aligned groups, and back.
//
// Replaces the Triton kernels of torchao/prototype/moe_training/ep/: `generate_permute_indices` (kernels.py:132-214; semantics of
// its own CPU restatement fill_indices_cpu :94-129), the row gather `x_padded[permuted_indices]` of permute_and_pad /
// _PermuteMXFP8FwdHPBwd.forward (permute.py:60-125, 170-204) and the row scatter `out[permuted_indices] = y` of
// _UnpermuteHPFwdMXFP8Bwd.forward / _unpermute_bf16 (unpermute.py:23-47, 140-158).
//
// After the all-to-all a rank holds its tokens rank-major: for source rank r, the tokens of local expert 0, then expert 1, ...
// (counts tokens_per_expert_group[r * E + e]).  The grouped GEMM wants them expert-major with every expert's group padded to a
// multiple of `alignment` rows (empty experts get one aligned block):
//   m_sizes[e]   = align_up(max(sum_r count[r, e], alignment))        m_offsets = cumsum(m_sizes)
//   permuted_indices[m_offsets[e] - m_sizes[e] + j] = start[r, e] + (j - sum_{r' < r} count[r', e])   for the j-th token of expert e
//   every other position = -1 (the reference gathers row -1 = a zero row it appends to x)
// All of it is index arithmetic over R * E counters plus HBM-bound row copies (2 B moved per payload byte): one wave per row,
// 16-byte lanes, four loads in flight; the counters are read from L2 by every wave.
#include "common.h"

namespace ao {
namespace {

constexpr int kRowsPerWg = 4;  // one wave per row

// one workgroup: start[r * E + e] (exclusive prefix sum of the counts in rank-major order), m_sizes, m_offsets
__global__ __launch_bounds__(256) void moe_permute_sizes_kernel(const int32_t* __restrict__ counts, int32_t* __restrict__ start,
                                                                int32_t* __restrict__ m_sizes, int32_t* __restrict__ m_offsets, int E, int R,
                                                                int alignment) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = E * R;
  // exclusive scan of counts[0 .. n) by wave 0, 64 per pass
  if (tid < 64) {
    int carry = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const int v = i < n ? counts[i] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      if (i < n) start[i] = carry + incl - v;
      carry += __shfl(incl, 63);
    }
    // aligned group sizes and their inclusive scan, same wave
    int off = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + lane;
      int size = 0;
      if (e < E) {
        int tot = 0;
        for (int r = 0; r < R; ++r) tot += counts[r * E + e];
        tot = max(tot, alignment);
        size = ((tot + alignment - 1) / alignment) * alignment;
      }
      int incl = size;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      if (e < E) { m_sizes[e] = size; m_offsets[e] = off + incl; }
      off += __shfl(incl, 63);
    }
  }
}

// one thread per output position
__global__ __launch_bounds__(256) void moe_permute_fill_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ start,
                                                               const int32_t* __restrict__ m_sizes, const int32_t* __restrict__ m_offsets,
                                                               int32_t* __restrict__ out, int E, int R, int max_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_len) return;
  // expert = first e with i < m_offsets[e] (binary search; E is small)
  int lo = 0, hi = E;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (i < m_offsets[mid]) hi = mid; else lo = mid + 1;
  }
  int v = -1;
  if (lo < E) {
    int j = i - (m_offsets[lo] - m_sizes[lo]);
    for (int r = 0; r < R; ++r) {
      const int c = counts[r * E + lo];
      if (j < c) { v = start[r * E + lo] + j; break; }
      j -= c;
    }
  }
  out[i] = v;
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

// out[i] = idx[i] in [0, T) ? in[idx[i]] : 0      (index -1 / T = the reference's appended zero row)
template <typename V>
__global__ __launch_bounds__(64 * kRowsPerWg) void moe_gather_rows_kernel(const V* __restrict__ in, const int32_t* __restrict__ idx,
                                                                          V* __restrict__ out, int64_t T, int64_t L, int64_t nvec) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerWg + (threadIdx.x >> 6);
  if (row >= L) return;
  V* dst = out + row * nvec;
  const int64_t j = idx[row];
  if (j >= 0 && j < T) {
    copy_row(in + j * nvec, dst, nvec, lane);
  } else {
    V z;
    __builtin_memset(&z, 0, sizeof(V));
    for (int64_t i = lane; i < nvec; i += 64) __builtin_nontemporal_store(z, dst + i);
  }
}

// out[idx[i]] = in[i] for idx[i] in [0, T)      (rows sent to the dummy row are dropped)
template <typename V>
__global__ __launch_bounds__(64 * kRowsPerWg) void moe_scatter_rows_kernel(const V* __restrict__ in, const int32_t* __restrict__ idx,
                                                                           V* __restrict__ out, int64_t T, int64_t L, int64_t nvec) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerWg + (threadIdx.x >> 6);
  if (row >= L) return;
  const int64_t j = idx[row];
  if (j >= 0 && j < T) copy_row(in + row * nvec, out + j * nvec, nvec, lane);
}

template <bool GATHER>
int launch_rows(const void* in, const int32_t* idx, void* out, int64_t T, int64_t L, int64_t row_bytes, hipStream_t st) {
  const dim3 grid((unsigned)((L + kRowsPerWg - 1) / kRowsPerWg)), block(64 * kRowsPerWg);
  const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  auto go = [&](auto tag) {
    using V = decltype(tag);
    const int64_t nvec = row_bytes / (int64_t)sizeof(V);
    if constexpr (GATHER)
      ao::launch(moe_gather_rows_kernel<V>, grid, block, 0, st, static_cast<const V*>(in), idx, static_cast<V*>(out), T, L, nvec);
    else
      ao::launch(moe_scatter_rows_kernel<V>, grid, block, 0, st, static_cast<const V*>(in), idx, static_cast<V*>(out), T, L, nvec);
  };
  if (a16) go(u32x4{});
  else if (row_bytes % 4 == 0) go(uint32_t{});
  else if (row_bytes % 2 == 0) go(uint16_t{});
  else go(uint8_t{});
  AO_LAUNCH_CHECK(GATHER ? "moe_gather_rows_kernel launch" : "moe_scatter_rows_kernel launch");
  return AO_OK;
}

}  // namespace
}  // namespace ao

using namespace ao;

extern "C" int ao_moe_permute_indices(const int32_t* tokens_per_expert_group, int32_t* start_workspace, int32_t* permuted_indices,
                                      int32_t* m_sizes, int32_t* m_offsets, int64_t experts_per_rank, int64_t num_ranks, int64_t max_len,
                                      int alignment, void* stream) {
  AO_REQUIRE_PTR(tokens_per_expert_group);
  AO_REQUIRE_PTR(start_workspace);
  AO_REQUIRE_PTR(permuted_indices);
  AO_REQUIRE_PTR(m_sizes);
  AO_REQUIRE_PTR(m_offsets);
  AO_REQUIRE(experts_per_rank > 0 && num_ranks > 0 && experts_per_rank * num_ranks < (1 << 24), "ao_moe_permute_indices: bad E=%lld R=%lld",
             (long long)experts_per_rank, (long long)num_ranks);
  AO_REQUIRE(alignment > 0, "ao_moe_permute_indices: alignment must be positive, got %d", alignment);
  AO_REQUIRE(max_len >= 0 && max_len < (1ll << 31), "ao_moe_permute_indices: bad max_len=%lld", (long long)max_len);
  hipStream_t st = static_cast<hipStream_t>(stream);
  ao::launch(moe_permute_sizes_kernel, dim3(1), dim3(256), 0, st, tokens_per_expert_group, start_workspace, m_sizes, m_offsets,
             (int)experts_per_rank, (int)num_ranks, alignment);
  AO_LAUNCH_CHECK("moe_permute_sizes_kernel launch");
  if (max_len > 0) {
    ao::launch(moe_permute_fill_kernel, dim3((unsigned)((max_len + 255) / 256)), dim3(256), 0, st, tokens_per_expert_group, start_workspace,
               m_sizes, m_offsets, permuted_indices, (int)experts_per_rank, (int)num_ranks, (int)max_len);
    AO_LAUNCH_CHECK("moe_permute_fill_kernel launch");
  }
  return AO_OK;
}

extern "C" int ao_moe_gather_rows(const void* inputs, const int32_t* indices, void* out, int64_t num_rows_in, int64_t num_rows_out,
                                  int64_t row_bytes, void* stream) {
  AO_REQUIRE(num_rows_in >= 0 && num_rows_out >= 0 && row_bytes > 0, "ao_moe_gather_rows: bad shape in=%lld out=%lld row_bytes=%lld",
             (long long)num_rows_in, (long long)num_rows_out, (long long)row_bytes);
  if (num_rows_out == 0) return AO_OK;
  AO_REQUIRE_PTR(indices);
  AO_REQUIRE_PTR(out);
  if (num_rows_in > 0) AO_REQUIRE_PTR(inputs);
  return launch_rows<true>(inputs, indices, out, num_rows_in, num_rows_out, row_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int ao_moe_scatter_rows(const void* inputs, const int32_t* indices, void* out, int64_t num_rows_in, int64_t num_rows_out,
                                   int64_t row_bytes, void* stream) {
  AO_REQUIRE(num_rows_in >= 0 && num_rows_out >= 0 && row_bytes > 0, "ao_moe_scatter_rows: bad shape in=%lld out=%lld row_bytes=%lld",
             (long long)num_rows_in, (long long)num_rows_out, (long long)row_bytes);
  if (num_rows_in == 0 || num_rows_out == 0) return AO_OK;
  AO_REQUIRE_PTR(inputs);
  AO_REQUIRE_PTR(indices);
  AO_REQUIRE_PTR(out);
  return launch_rows<false>(inputs, indices, out, num_rows_out, num_rows_in, row_bytes, static_cast<hipStream_t>(stream));
}
