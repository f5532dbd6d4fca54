// torchao's own op names on MI355X, registered the way the reference registers them: through PyTorch's STABLE ABI
// (STABLE_TORCH_LIBRARY_IMPL + TORCH_BOX over torch::stable::Tensor: torchao/csrc/cuda/mx_kernels/mxfp8_extension.cpp:1-16, 425-430), so that
// this object -- like torchao's _C*.so -- does not depend on the C++ ABI of the ATen headers it was compiled against:
//
//   torchao::mxfp8_quantize(Tensor input, bool rowwise, bool colwise, int scale_dim_x, int scale_dim_y, str fp8_format, str scaling_mode)
//                                                                       -> (Tensor, Tensor, Tensor, Tensor)   mxfp8_extension.cpp:109-188
//   torchao::fused_pad_token_groups(Tensor inputs, Tensor group_offsets, int alignment_size) -> (Tensor, Tensor, Tensor)      :226-300
//   torchao::fused_unpad_token_groups(Tensor inputs, Tensor group_offsets, Tensor padded_group_start_offsets, int num_tokens,
//                                     int alignment_size) -> Tensor                                                           :302-420
//
// Schemas live in Python (torchao's own, or ao_amd/torch_ops.py when torchao is not imported); dispatch key CUDA (= HIP on ROCm).
// Host-only C++ over the C ABI of include/ao_mi355.h.  Ownership / errors as in the reference: inputs borrowed, outputs
// allocated with torch::stable::new_empty and returned, the current stream from the AOTI shim, a device guard per call, failures through
// STD_TORCH_CHECK (a C++ exception -> Python RuntimeError), never exit().
//   torchao::mx_block_rearrange_2d_M_groups(Tensor scales_tensor, Tensor input_offsets, int chunks_per_tb) -> Tensor                :178-300
//     (round 6: the cuBLAS 128 x 4 blocked swizzle of the scales as a data-format op; the GEMMs here take row-major E8M0 scales)
// The aten:: overrides (opt-in) and the ao_mi355_c:: test namespace stay in binding.cpp: they take ATen-only argument kinds
// (ScalarType?, Tensor? with defaults) for schemas PyTorch core owns.
#include <torch/csrc/inductor/aoti_torch/c/shim.h>
#include <torch/csrc/stable/accelerator.h>
#include <torch/csrc/stable/library.h>
#include <torch/csrc/stable/ops.h>
#include <torch/csrc/stable/tensor.h>
#include <torch/headeronly/core/ScalarType.h>
#include <torch/headeronly/util/Exception.h>
#include <torch/headeronly/util/shim_utils.h>

#include <cstdint>
#include <string>
#include <tuple>

#include "../../include/ao_mi355.h"

namespace {

using torch::stable::Tensor;
using torch::headeronly::ScalarType;
namespace tsa = torch::stable::accelerator;

void* current_stream(const Tensor& t) {
  void* s = nullptr;
  TORCH_ERROR_CODE_CHECK(aoti_torch_get_current_cuda_stream(t.get_device_index(), &s));
  return s;
}

#define AO_RC(call, op) STD_TORCH_CHECK((call) == AO_OK, op, ": ", ao_last_error())

// A new uninitialised tensor on `like`'s device with explicit strides, through the C shim.  torch 2.10's stable ScalarType conversion has
// no float8_e8m0fnu yet (stableivalue_conversions.h: "Not yet supported ScalarType"; the reference needs torch >= 2.11 for its new_empty
// of that dtype, mxfp8_extension.cpp:146): the shim takes the dtype as the enum's integer, which it is on every version.
Tensor empty_strided_like(const Tensor& like, std::initializer_list<int64_t> sizes, std::initializer_list<int64_t> strides, ScalarType dtype) {
  int32_t device_type = 0;
  TORCH_ERROR_CODE_CHECK(aoti_torch_get_device_type(like.get(), &device_type));
  AtenTensorHandle h = nullptr;
  TORCH_ERROR_CODE_CHECK(aoti_torch_empty_strided((int64_t)sizes.size(), sizes.begin(), strides.begin(), static_cast<int32_t>(dtype), device_type,
                                                  like.get_device_index(), &h));
  return Tensor(h);
}

void check_gpu(const Tensor& t, const char* op, const char* name) {
  STD_TORCH_CHECK(t.is_cuda(), op, ": ", name, " must be on the GPU (the MI355X backend has no CPU fallback)");
}

int scaling_mode_of(const std::string& s, const char* op) {
  if (s == "floor") return AO_MX_SCALE_FLOOR;
  if (s == "rceil") return AO_MX_SCALE_RCEIL;
  STD_TORCH_CHECK(false, op, ": scaling_mode must be 'floor' or 'rceil', got: ", s);
  return 0;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> mxfp8_quantize(const Tensor& input, bool rowwise, bool colwise, int64_t scale_dim_x, int64_t scale_dim_y,
                                                          std::string fp8_format, std::string scaling_mode) {
  const char* op = "mxfp8_quantize";
  check_gpu(input, op, "input");
  STD_TORCH_CHECK(input.is_contiguous(), op, ": input must be contiguous");
  STD_TORCH_CHECK(input.dim() == 2, op, ": input must be 2D");
  STD_TORCH_CHECK(input.scalar_type() == ScalarType::BFloat16, op,
                  ": input must be bfloat16 on MI355X (the float32 flavour of the reference is not on the inference path)");
  STD_TORCH_CHECK(rowwise || colwise, op, ": At least one of rowwise or colwise must be true");
  STD_TORCH_CHECK(scale_dim_x == 1 || scale_dim_x == 32, op, ": scale_dim_x must be 1 or 32, got: ", scale_dim_x);
  STD_TORCH_CHECK(scale_dim_y == 1 || scale_dim_y == 32, op, ": scale_dim_y must be 1 or 32, got: ", scale_dim_y);
  STD_TORCH_CHECK(fp8_format == "e4m3", op, ": fp8_format must be 'e4m3', got: ", fp8_format);
  STD_TORCH_CHECK(!rowwise || scale_dim_x == 32, op, ": rowwise output requires scale_dim_x == 32");
  STD_TORCH_CHECK(!colwise || scale_dim_y == 32, op, ": colwise output requires scale_dim_y == 32");
  const int mode = scaling_mode_of(scaling_mode, op);
  const int64_t rows = input.size(0), cols = input.size(1);
  STD_TORCH_CHECK(rows >= 32 && rows % 32 == 0, op, ": rows must be a multiple of 32");
  STD_TORCH_CHECK(cols >= 32 && cols % 32 == 0, op, ": cols must be a multiple of 32");
  tsa::DeviceGuard guard(input.get_device_index());
  const auto f8 = ScalarType::Float8_e4m3fn, e8 = ScalarType::Float8_e8m0fnu;
  const uint16_t* x = reinterpret_cast<const uint16_t*>(input.data_ptr());
  Tensor out_r = empty_strided_like(input, {0}, {1}, f8), sc_r = empty_strided_like(input, {0}, {1}, e8);
  Tensor out_c = empty_strided_like(input, {0}, {1}, f8), sc_c = empty_strided_like(input, {0}, {1}, e8);
  if (rowwise) {
    out_r = empty_strided_like(input, {rows, cols}, {cols, 1}, f8);
    sc_r = empty_strided_like(input, {rows, cols / 32}, {cols / 32, 1}, e8);
    AO_RC(ao_mxfp8_quantize_rowwise(x, reinterpret_cast<uint8_t*>(out_r.data_ptr()), reinterpret_cast<uint8_t*>(sc_r.data_ptr()), rows, cols, mode,
                                    current_stream(input)), op);
  }
  if (colwise) {
    // column-major data {rows, cols} with strides {1, rows}; scales {cols, rows / 32} with strides {1, cols} (mxfp8_extension.cpp:147-158):
    // the kernel writes [cols][rows] and [rows / 32][cols] row-major, which is that memory
    out_c = empty_strided_like(input, {rows, cols}, {1, rows}, f8);
    sc_c = empty_strided_like(input, {cols, rows / 32}, {1, cols}, e8);
    AO_RC(ao_mxfp8_quantize_colwise(x, reinterpret_cast<uint8_t*>(out_c.data_ptr()), reinterpret_cast<uint8_t*>(sc_c.data_ptr()), rows, cols, mode,
                                    current_stream(input)), op);
  }
  return std::make_tuple(out_r, out_c, sc_r, sc_c);
}

int elem_bytes_of(const Tensor& t, const char* op) {
  STD_TORCH_CHECK(t.scalar_type() == ScalarType::BFloat16 || t.scalar_type() == ScalarType::Float, op, ": inputs must be bfloat16 or float32");
  return t.scalar_type() == ScalarType::Float ? 4 : 2;
}

std::tuple<Tensor, Tensor, Tensor> fused_pad_token_groups(const Tensor& inputs, const Tensor& offsets, int64_t alignment_size) {
  const char* op = "fused_pad_token_groups";
  check_gpu(inputs, op, "inputs");
  check_gpu(offsets, op, "group_offsets");
  STD_TORCH_CHECK(inputs.dim() == 2 && inputs.is_contiguous(), op, ": inputs must be a contiguous 2-D tensor");
  STD_TORCH_CHECK(offsets.dim() == 1 && offsets.scalar_type() == ScalarType::Int && offsets.is_contiguous(), op, ": group_offsets must be int32 [num_groups]");
  const int eb = elem_bytes_of(inputs, op);
  const int64_t T = inputs.size(0), D = inputs.size(1), G = offsets.size(0);
  const int64_t rows = ao_moe_padded_rows(T, G, (int)alignment_size);
  STD_TORCH_CHECK(rows >= 0, op, ": ", ao_last_error());
  tsa::DeviceGuard guard(inputs.get_device_index());
  Tensor padded = torch::stable::new_empty(inputs, {rows, D});
  Tensor starts = torch::stable::new_empty(offsets, {G}), ends = torch::stable::new_empty(offsets, {G});
  AO_RC(ao_moe_pad_token_groups(inputs.data_ptr(), reinterpret_cast<const int32_t*>(offsets.data_ptr()), padded.data_ptr(),
                                reinterpret_cast<int32_t*>(starts.data_ptr()), reinterpret_cast<int32_t*>(ends.data_ptr()), T, D, eb, G,
                                (int)alignment_size, current_stream(inputs)), op);
  return std::make_tuple(padded, starts, ends);
}

Tensor fused_unpad_token_groups(const Tensor& padded, const Tensor& offsets, const Tensor& padded_starts, int64_t num_tokens, int64_t alignment_size) {
  const char* op = "fused_unpad_token_groups";
  (void)alignment_size;
  check_gpu(padded, op, "inputs");
  check_gpu(offsets, op, "group_offsets");
  check_gpu(padded_starts, op, "padded_group_start_offsets");
  STD_TORCH_CHECK(padded.dim() == 2 && padded.is_contiguous(), op, ": inputs must be a contiguous 2-D tensor");
  STD_TORCH_CHECK(offsets.scalar_type() == ScalarType::Int && padded_starts.scalar_type() == ScalarType::Int && offsets.dim() == 1 &&
                      padded_starts.dim() == 1 && offsets.size(0) == padded_starts.size(0),
                  op, ": offsets must be int32 tensors of the same shape");
  STD_TORCH_CHECK(num_tokens >= 0, op, ": num_tokens must be non-negative");
  const int eb = elem_bytes_of(padded, op);
  tsa::DeviceGuard guard(padded.get_device_index());
  const Tensor of = torch::stable::contiguous(offsets), ps = torch::stable::contiguous(padded_starts);
  Tensor out = torch::stable::new_empty(padded, {num_tokens, padded.size(1)});
  AO_RC(ao_moe_unpad_token_groups(padded.data_ptr(), reinterpret_cast<const int32_t*>(of.data_ptr()), reinterpret_cast<const int32_t*>(ps.data_ptr()),
                                  out.data_ptr(), num_tokens, padded.size(1), eb, offsets.size(0), current_stream(padded)), op);
  return out;
}

// mxfp8_extension.cpp:178-300: same checks, same output shape (rows + 128 groups, 4 ceil(cols / 4)); the reference's limit of 32 groups
// (a shared-memory table of its kernel) does not apply; chunks_per_tb is validated and otherwise its launch-shape knob
Tensor mx_block_rearrange_2d_M_groups(const Tensor& scales, const Tensor& offsets, int64_t chunks_per_tb) {
  const char* op = "mx_block_rearrange_2d_M_groups";
  check_gpu(scales, op, "scales_tensor");
  check_gpu(offsets, op, "input_group_end_offsets");
  STD_TORCH_CHECK(scales.dim() == 2, "scales_tensor must be 2D");
  STD_TORCH_CHECK(scales.is_contiguous(), "scales_tensor must be contiguous (row-major)");
  // (the dtype as the shim's integer: torch 2.10's stable scalar_type() cannot name float8_e8m0fnu yet, see empty_strided_like)
  int32_t dt = 0;
  TORCH_ERROR_CODE_CHECK(aoti_torch_get_dtype(scales.get(), &dt));
  STD_TORCH_CHECK(dt == static_cast<int32_t>(ScalarType::Byte) || dt == static_cast<int32_t>(ScalarType::Float8_e8m0fnu), "scales_tensor must be uint8 or e8m0");
  STD_TORCH_CHECK(offsets.scalar_type() == ScalarType::Int, "input_group_end_offsets must be int32");
  STD_TORCH_CHECK(offsets.dim() == 1 && offsets.is_contiguous(), "input_group_end_offsets must be 1D");
  STD_TORCH_CHECK(chunks_per_tb == 1 || chunks_per_tb == 4 || chunks_per_tb == 8 || chunks_per_tb == 16, "chunks_per_tb must be 1, 4, 8, or 16, got: ",
                  chunks_per_tb);
  const int64_t rows = scales.size(0), cols = scales.size(1), G = offsets.size(0);
  STD_TORCH_CHECK(G > 0, op, ": input_group_end_offsets must not be empty");
  tsa::DeviceGuard guard(scales.get_device_index());
  const int64_t out_rows = ao_mx_blocked_rows(rows, G), pcols = (cols + 3) / 4 * 4;
  Tensor out = empty_strided_like(scales, {out_rows, pcols}, {pcols, 1}, static_cast<ScalarType>(dt));  // (every byte is written by the kernel)
  AO_RC(ao_mx_block_rearrange_2d_m_groups(reinterpret_cast<const uint8_t*>(scales.data_ptr()), reinterpret_cast<const int32_t*>(offsets.data_ptr()),
                                          reinterpret_cast<uint8_t*>(out.data_ptr()), rows, cols, G, current_stream(scales)), op);
  return out;
}

}  // namespace

STABLE_TORCH_LIBRARY_IMPL(torchao, CUDA, m) {
  m.impl("mx_block_rearrange_2d_M_groups", TORCH_BOX(&mx_block_rearrange_2d_M_groups));
  m.impl("mxfp8_quantize", TORCH_BOX(&mxfp8_quantize));
  m.impl("fused_pad_token_groups", TORCH_BOX(&fused_pad_token_groups));
  m.impl("fused_unpad_token_groups", TORCH_BOX(&fused_unpad_token_groups));
}
