// Dispatcher registration of the MI355X kernels, compiled into ao_amd/_C_mi355_ops.so: the boundary torchao's own loader
// uses (torchao/__init__.py:89-94 torch.ops.load_library()s every torchao/_C*.so, whose TORCH_LIBRARY_IMPL blocks register
// device kernels for schemas defined in Python -- csrc/cuda/mx_kernels/mxfp8_extension.cpp:425-430).  This file is that for
// gfx950: host-only C++ over the C ABI of include/ao_mi355.h (no device code here), dispatch key CUDA (= HIP on ROCm).
//
//   * (round 6) torchao::mxfp8_quantize, torchao::fused_pad_token_groups, torchao::fused_unpad_token_groups -- the reference's own op
//     names -- are registered from binding_stable.cpp through STABLE_TORCH_LIBRARY_IMPL / TORCH_BOX, as the reference does; this file
//     keeps what needs ATen argument kinds.
//   * TORCH_LIBRARY_IMPL(aten, CUDA), only when AO_MI355_OVERRIDE_ATEN=1 is set when the library is loaded:
//     aten::_weight_int4pack_mm, aten::_convert_weight_to_int4pack (int4_tile_packed_to_4d_tensor.py:202,287),
//     aten::_int_mm (int8/kernels.py:38-40,70), aten::_scaled_mm with rowwise or tensorwise scales (float8/inference.py:104-123),
//     aten::_scaled_grouped_mm with MXFP8 operands (mxfp8_grouped_mm.py:541) -- so that an UNMODIFIED torchao's
//     Int4TilePackedTo4dTensor / Int8Tensor / Float8Tensor reach these kernels through PyTorch-ROCm's dispatcher.
//     Variants of those ops outside the low-bit path (blockwise scales, fp16 outputs, ...) raise instead of silently
//     computing something else: the override is opt-in for exactly that reason.
//   * TORCH_LIBRARY(ao_mi355_c, ...): the same kernels under their own namespace, always registered (tests, opcheck).
//
// Ownership / errors follow the reference: inputs borrowed, outputs allocated and returned, current
// stream, device guard per call, failures become c10::Error (Python RuntimeError), never exit().
#include <cstdlib>
#include <string>
#include <tuple>

#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include "../../include/ao_mi355.h"

namespace {

using at::Tensor;

// PyTorch-ROCm presents HIP devices as device type "cuda": the stream comes from the masquerading accessor, the device guard
// is the generic one (it resolves to the masquerading guard implementation registered for that device type)
void* current_stream(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }

void check_rc(int rc, const char* op) { TORCH_CHECK(rc == AO_OK, op, ": ", ao_last_error()); }

void check_gpu(const Tensor& t, const char* op, const char* name) {
  TORCH_CHECK(t.is_cuda(), op, ": ", name, " must be on the GPU (the MI355X backend has no CPU fallback)");
}

// ---- int4 -------------------------------------------------------------------------------------------------------
Tensor weight_int4pack_mm(const Tensor& self, const Tensor& mat2, int64_t qGroupSize, const Tensor& qScaleAndZeros) {
  const char* op = "_weight_int4pack_mm";
  check_gpu(self, op, "self"); check_gpu(mat2, op, "mat2"); check_gpu(qScaleAndZeros, op, "qScaleAndZeros");
  TORCH_CHECK(self.dim() == 2 && self.scalar_type() == at::kBFloat16, op, ": self must be a 2-D bfloat16 tensor");
  TORCH_CHECK(mat2.dim() == 4 && mat2.scalar_type() == at::kInt && mat2.size(2) == 32 && mat2.size(3) == 4,
              op, ": mat2 must be the int32 [N/8, K/128, 32, 4] tile-packed weight (innerKTiles = 8)");
  TORCH_CHECK(qScaleAndZeros.dim() == 3 && qScaleAndZeros.scalar_type() == at::kBFloat16 && qScaleAndZeros.size(2) == 2,
              op, ": qScaleAndZeros must be bfloat16 [K/g, N, 2]");
  const int64_t M = self.size(0), K = self.size(1), N = mat2.size(0) * 8;
  TORCH_CHECK(mat2.size(1) * 128 == K, op, ": K mismatch between self and mat2");
  TORCH_CHECK(qScaleAndZeros.size(1) == N && qScaleAndZeros.size(0) * qGroupSize == K, op, ": qScaleAndZeros shape does not match");
  c10::DeviceGuard guard(self.device());
  const Tensor x = self.contiguous(), q = mat2.contiguous(), sz = qScaleAndZeros.contiguous();
  Tensor y = at::empty({M, N}, self.options());
  check_rc(ao_int4_weight_int4pack_mm(reinterpret_cast<const uint16_t*>(x.data_ptr()), q.data_ptr<int32_t>(),
                                      reinterpret_cast<const uint16_t*>(sz.data_ptr()), reinterpret_cast<uint16_t*>(y.data_ptr()), M, N, K,
                                      (int)qGroupSize, current_stream(self)), op);
  return y;
}

Tensor convert_weight_to_int4pack(const Tensor& self, int64_t innerKTiles) {
  const char* op = "_convert_weight_to_int4pack";
  check_gpu(self, op, "self");
  TORCH_CHECK(self.dim() == 2 && self.scalar_type() == at::kByte, op, ": self must be a 2-D uint8 tensor [N, K/2]");
  TORCH_CHECK(innerKTiles == 8, op, ": innerKTiles must be 8 on MI355X (torchao fixes it), got ", innerKTiles);
  const int64_t N = self.size(0), K = self.size(1) * 2;
  c10::DeviceGuard guard(self.device());
  const Tensor w = self.contiguous();
  Tensor q = at::empty({N / 8, K / 128, 32, 4}, self.options().dtype(at::kInt));
  check_rc(ao_int4_convert_weight_to_int4pack(w.data_ptr<uint8_t>(), q.data_ptr<int32_t>(), N, K, 8, current_stream(self)), op);
  return q;
}

// ---- int8 -------------------------------------------------------------------------------------------------------
Tensor int_mm(const Tensor& self, const Tensor& mat2) {
  const char* op = "_int_mm";
  check_gpu(self, op, "self"); check_gpu(mat2, op, "mat2");
  TORCH_CHECK(self.dim() == 2 && mat2.dim() == 2 && self.scalar_type() == at::kChar && mat2.scalar_type() == at::kChar,
              op, ": expected 2-D int8 tensors");
  TORCH_CHECK(self.size(1) == mat2.size(0), op, ": shapes cannot be multiplied");
  c10::DeviceGuard guard(self.device());
  const Tensor a = self.contiguous(), bt = mat2.t().contiguous();  // K-major weight: free for the reference's `.contiguous().t()`
  const int64_t M = a.size(0), K = a.size(1), N = bt.size(0);
  Tensor c = at::empty({M, N}, self.options().dtype(at::kInt));
  check_rc(ao_int8_int_mm(a.data_ptr<int8_t>(), bt.data_ptr<int8_t>(), c.data_ptr<int32_t>(), M, N, K, current_stream(self)), op);
  return c;
}

// ---- fp8 rowwise ------------------------------------------------------------------------------------------------
Tensor scaled_mm(const Tensor& self, const Tensor& mat2, const Tensor& scale_a, const Tensor& scale_b,
                 const std::optional<Tensor>& bias, const std::optional<Tensor>& scale_result,
                 std::optional<c10::ScalarType> out_dtype, bool use_fast_accum) {
  const char* op = "_scaled_mm (MI355X e4m3)";
  (void)use_fast_accum;  // fp32 accumulation on the scaled MFMA either way
  check_gpu(self, op, "self"); check_gpu(mat2, op, "mat2");
  TORCH_CHECK(self.scalar_type() == at::kFloat8_e4m3fn && mat2.scalar_type() == at::kFloat8_e4m3fn,
              op, ": only float8_e4m3fn operands are implemented (OCP e4m3 is gfx950's fp8; unset AO_MI355_OVERRIDE_ATEN for other dtypes)");
  TORCH_CHECK(self.dim() == 2 && mat2.dim() == 2 && self.size(1) == mat2.size(0), op, ": shapes cannot be multiplied");
  TORCH_CHECK(!scale_result.has_value(), op, ": scale_result is not implemented");
  TORCH_CHECK(!out_dtype.has_value() || *out_dtype == at::kBFloat16, op, ": only bfloat16 outputs are implemented");
  const int64_t M = self.size(0), K = self.size(1), N = mat2.size(1);
  TORCH_CHECK(scale_a.scalar_type() == at::kFloat && scale_b.scalar_type() == at::kFloat, op, ": scales must be float32");
  const bool rowwise = scale_a.numel() == M && scale_b.numel() == N;
  const bool tensorwise = scale_a.numel() == 1 && scale_b.numel() == 1;
  TORCH_CHECK(rowwise || tensorwise, op, ": scales must be rowwise (scale_a [M,1], scale_b [1,N]) or tensorwise ([1,1] both); blockwise is not implemented");
  c10::DeviceGuard guard(self.device());
  const Tensor a = self.contiguous(), bt = mat2.t().contiguous();  // mat2 is column-major [K,N] = row-major [N,K]: no copy
  // tensorwise: the same epilogue with the two scalars broadcast over rows / columns (M + N floats)
  const Tensor sa = (rowwise ? scale_a.reshape({-1}) : scale_a.reshape({1}).expand({M})).contiguous();
  const Tensor sb = (rowwise ? scale_b.reshape({-1}) : scale_b.reshape({1}).expand({N})).contiguous();
  Tensor bb;
  if (bias.has_value() && bias->defined()) {
    TORCH_CHECK(bias->numel() == N, op, ": bias must have N elements");
    bb = bias->to(at::kBFloat16).contiguous();
  }
  Tensor y = at::empty({M, N}, self.options().dtype(at::kBFloat16));
  check_rc(ao_fp8_scaled_mm(reinterpret_cast<const uint8_t*>(a.data_ptr()), reinterpret_cast<const uint8_t*>(bt.data_ptr()),
                            sa.data_ptr<float>(), sb.data_ptr<float>(), bb.defined() ? reinterpret_cast<const uint16_t*>(bb.data_ptr()) : nullptr,
                            reinterpret_cast<uint16_t*>(y.data_ptr()), M, N, K, current_stream(self)), op);
  return y;
}

// ---- MXFP8 ------------------------------------------------------------------------------------------------------
// aten::_scaled_grouped_mm(self, mat2, scale_a, scale_b, offs?, bias?, scale_result?, out_dtype?, use_fast_accum)
// MXFP8 2d-3d form (mxfp8_grouped_mm.py:541): self e4m3 [M, K]; mat2 e4m3 [E, K, N] whose experts are K-major (the
// transpose of [E, N, K]); scale_a e8m0 [M, K/32]; scale_b e8m0 [E, N, K/32] -- plain row-major scales: CDNA4's scaled
// MFMA takes them from VGPRs, the cuBLAS 128x4 blocked swizzle (mx_block_rearrange_2d_M_groups, binding_stable.cpp) has no role on this path.
Tensor scaled_grouped_mm(const Tensor& self, const Tensor& mat2, const Tensor& scale_a, const Tensor& scale_b,
                         const std::optional<Tensor>& offs, const std::optional<Tensor>& bias, const std::optional<Tensor>& scale_result,
                         std::optional<c10::ScalarType> out_dtype, bool use_fast_accum) {
  const char* op = "_scaled_grouped_mm (MI355X MXFP8)";
  (void)use_fast_accum;
  check_gpu(self, op, "self"); check_gpu(mat2, op, "mat2");
  TORCH_CHECK(self.scalar_type() == at::kFloat8_e4m3fn && mat2.scalar_type() == at::kFloat8_e4m3fn, op, ": float8_e4m3fn operands expected");
  TORCH_CHECK(self.dim() == 2 && mat2.dim() == 3 && self.size(1) == mat2.size(1), op, ": expected self [M, K] and mat2 [E, K, N]");
  TORCH_CHECK(offs.has_value() && offs->defined() && offs->scalar_type() == at::kInt && offs->dim() == 1 && offs->size(0) == mat2.size(0),
              op, ": offs must be int32 [E] (cumulative group ends)");
  TORCH_CHECK(!(bias.has_value() && bias->defined()) && !scale_result.has_value(), op, ": bias / scale_result are not implemented");
  TORCH_CHECK(!out_dtype.has_value() || *out_dtype == at::kBFloat16, op, ": only bfloat16 outputs are implemented");
  const int64_t M = self.size(0), K = self.size(1), E = mat2.size(0), N = mat2.size(2);
  const bool e8 = (scale_a.scalar_type() == at::kFloat8_e8m0fnu || scale_a.scalar_type() == at::kByte) &&
                  (scale_b.scalar_type() == at::kFloat8_e8m0fnu || scale_b.scalar_type() == at::kByte);
  TORCH_CHECK(e8 && scale_a.numel() == M * (K / 32) && scale_b.numel() == E * N * (K / 32), op,
              ": only the MXFP8 form is implemented: E8M0 scales [M, K/32] and [E, N, K/32], row-major, not blocked");
  c10::DeviceGuard guard(self.device());
  const Tensor a = self.contiguous(), b = mat2.transpose(1, 2).contiguous();  // [E, N, K]: no copy for the reference's layout
  const Tensor sa = scale_a.contiguous(), sb = scale_b.contiguous(), of = offs->contiguous();
  Tensor y = at::empty({M, N}, self.options().dtype(at::kBFloat16));
  check_rc(ao_mxfp8_grouped_mm(reinterpret_cast<const uint8_t*>(a.data_ptr()), reinterpret_cast<const uint8_t*>(sa.data_ptr()),
                               reinterpret_cast<const uint8_t*>(b.data_ptr()), reinterpret_cast<const uint8_t*>(sb.data_ptr()),
                               of.data_ptr<int32_t>(), reinterpret_cast<uint16_t*>(y.data_ptr()), M, N, K, E, current_stream(self)), op);
  return y;
}

bool override_aten() {
  const char* e = std::getenv("AO_MI355_OVERRIDE_ATEN");
  return e != nullptr && e[0] != '\0' && e[0] != '0';
}

}  // namespace

// (torchao::mxfp8_quantize / fused_pad_token_groups / fused_unpad_token_groups / mx_block_rearrange_2d_M_groups: binding_stable.cpp, through the stable ABI like the reference)

// Own namespace: always there, same functions (tests / opcheck / explicit use without touching aten).
TORCH_LIBRARY(ao_mi355_c, m) {
  m.def("_weight_int4pack_mm(Tensor self, Tensor mat2, int qGroupSize, Tensor qScaleAndZeros) -> Tensor");
  m.def("_convert_weight_to_int4pack(Tensor self, int innerKTiles) -> Tensor");
  m.def("_int_mm(Tensor self, Tensor mat2) -> Tensor");
  m.def("_scaled_mm(Tensor self, Tensor mat2, Tensor scale_a, Tensor scale_b, Tensor? bias=None, Tensor? scale_result=None, "
        "ScalarType? out_dtype=None, bool use_fast_accum=False) -> Tensor");
  m.def("_scaled_grouped_mm(Tensor self, Tensor mat2, Tensor scale_a, Tensor scale_b, Tensor? offs=None, Tensor? bias=None, "
        "Tensor? scale_result=None, ScalarType? out_dtype=None, bool use_fast_accum=False) -> Tensor");
  m.def("aten_overrides_active() -> bool", []() { return override_aten(); });
}
TORCH_LIBRARY_IMPL(ao_mi355_c, CUDA, m) {
  m.impl("_weight_int4pack_mm", &weight_int4pack_mm);
  m.impl("_convert_weight_to_int4pack", &convert_weight_to_int4pack);
  m.impl("_int_mm", &int_mm);
  m.impl("_scaled_mm", &scaled_mm);
  m.impl("_scaled_grouped_mm", &scaled_grouped_mm);
}

// The ATen names torchao's subclasses call.  Opt-in: replaces PyTorch core's kernels for every caller in the process.
TORCH_LIBRARY_IMPL(aten, CUDA, m) {
  if (!override_aten()) return;
  m.impl("_weight_int4pack_mm", &weight_int4pack_mm);
  m.impl("_convert_weight_to_int4pack", &convert_weight_to_int4pack);
  m.impl("_int_mm", &int_mm);
  m.impl("_scaled_mm", &scaled_mm);
  m.impl("_scaled_grouped_mm", &scaled_grouped_mm);
}
