"""GPU: the dispatcher boundary -- C++ registrations of ao_amd/_C_mi355_ops.so under the reference's op names, opcheck of
the custom ops' fake kernels, torch.compile(fullgraph=True) through the quantized subclasses, and the opt-in ATen
overrides (AO_MI355_OVERRIDE_ATEN=1) in a fresh process."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, np_from_torch_bf16
from oracle import mx_ref as MX

pytestmark = pytest.mark.gpu

from ao_amd import ops, torch_ops  # noqa: E402

DEV = "cuda"


def _randn_bf16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def test_ops_library_is_loaded_and_registers_reference_names():
    assert torch_ops.load_ops_library(), "ao_amd/_C_mi355_ops.so missing: python -m ao_amd.build"
    for name in ("mxfp8_quantize", "fused_pad_token_groups", "fused_unpad_token_groups"):
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"torchao::{name}", "CUDA"), name
        dump = torch._C._dispatch_dump(f"torchao::{name}")
        # the CUDA-key kernel is the C++ one inside the .so (round 6: a BOXED kernel registered through the stable ABI, like the
        # reference's STABLE_TORCH_LIBRARY_IMPL(torchao, CUDA) in mxfp8_extension.cpp:425-430), not a Python lambda
        assert "binding_stable.cpp" in dump and "[ boxed ]" in dump.split("CUDA:")[1].splitlines()[0], dump
    assert not torch.ops.ao_mi355_c.aten_overrides_active()  # opt-in only


@pytest.mark.parametrize("mode", ["rceil", "floor"])
@pytest.mark.parametrize("r,c", [(32, 32), (128, 256), (160, 96), (512, 4096)])
def test_torchao_mxfp8_quantize_rowwise_and_colwise_vs_oracle(r, c, mode):
    """The reference's 7-argument / 4-output schema (prototype/mx_formats/kernels.py:1022-1026), both casts bit-exact
    against to_mx (colwise == to_mx(x.t()).t(), mxfp8_extension.cpp:160-175)."""
    x = _randn_bf16((r, c), r + c, 3.0)
    x[0, :32] = 0.0
    xd = x.to(DEV)
    out_r, out_c, sc_r, sc_c = torch.ops.torchao.mxfp8_quantize(xd, True, True, 32, 32, "e4m3", mode)
    m = MX.RCEIL if mode == "rceil" else MX.FLOOR
    xn = x.float().numpy()
    q_r, s_r = MX.to_mx(xn, m)
    assert np.array_equal(out_r.view(torch.uint8).cpu().numpy(), q_r) and np.array_equal(sc_r.view(torch.uint8).cpu().numpy(), s_r)
    q_c, s_c = MX.to_mx(np.ascontiguousarray(xn.T), m)  # [C, R], [C, R/32]
    assert tuple(out_c.shape) == (r, c) and out_c.stride() == (1, r)
    assert tuple(sc_c.shape) == (c, r // 32) and sc_c.stride() == (1, c)
    assert np.array_equal(out_c.t().contiguous().view(torch.uint8).cpu().numpy(), q_c)
    assert np.array_equal(sc_c.contiguous().view(torch.uint8).cpu().numpy(), s_c)
    # rowwise only: the colwise outputs are the reference's empty placeholders
    o2 = torch.ops.torchao.mxfp8_quantize(xd, True, False, 32, 1, "e4m3", mode)
    assert o2[1].numel() == 0 and o2[3].numel() == 0 and torch.equal(o2[0].view(torch.uint8), out_r.view(torch.uint8))
    with pytest.raises(RuntimeError, match="At least one of rowwise or colwise"):
        torch.ops.torchao.mxfp8_quantize(xd, False, False, 32, 32, "e4m3", mode)
    with pytest.raises(RuntimeError, match="fp8_format must be 'e4m3'"):
        torch.ops.torchao.mxfp8_quantize(xd, True, False, 32, 32, "e5m2", mode)


@pytest.mark.parametrize("mode", ["rceil", "floor"])
def test_mxfp8_quantize_3d_is_the_colwise_cast_per_expert(mode):
    """mxfp8_quantize_cuda_3d, (32, 1) scale blocks (quant.py:1413-1440): every expert's [N, K] matrix cast along N."""
    E, N, K = 3, 96, 160
    x = _randn_bf16((E, N, K), 17)
    x[1, :32, 5] = 0
    x[2, 32:64, 7] *= 1e4
    q, s = ops.mxfp8_quantize_3d(x.to(DEV), scaling_mode=mode)
    assert tuple(q.shape) == (E, N, K) and q.stride() == (N * K, 1, N) and tuple(s.shape) == (E, K, N // 32)
    for e in range(E):
        qo, so = MX.to_mx(x[e].t().contiguous().float().numpy(), MX.RCEIL if mode == "rceil" else MX.FLOOR)  # [K, N], blocks along N
        assert np.array_equal(q[e].t().contiguous().view(torch.uint8).cpu().numpy(), qo)
        assert np.array_equal(s[e].contiguous().view(torch.uint8).cpu().numpy(), so)


def test_torchao_pad_unpad_ops_match_python_wrappers():
    x = _randn_bf16((100, 256), 3).to(DEV)
    offs = torch.tensor([10, 10, 57, 100], dtype=torch.int32, device=DEV)
    a = torch.ops.torchao.fused_pad_token_groups(x, offs, 32)
    b = ops.fused_pad_token_groups(x, offs, 32)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    back = torch.ops.torchao.fused_unpad_token_groups(a[0], offs, a[1], 100, 32)
    assert torch.equal(back, x)


def test_torchao_block_rearrange_op_matches_python_wrapper():
    """torchao::mx_block_rearrange_2d_M_groups through the dispatcher (uint8 and float8_e8m0fnu scales) = the C-ABI wrapper; the reference's
    own argument checks (mxfp8_extension.cpp:183-197) raise RuntimeError."""
    s = torch.randint(0, 256, (300, 128), dtype=torch.uint8, generator=torch.Generator().manual_seed(4)).to(DEV)
    offs = torch.tensor([10, 10, 170, 300], dtype=torch.int32, device=DEV)
    a = torch.ops.torchao.mx_block_rearrange_2d_M_groups(s, offs, 4)
    assert tuple(a.shape) == (300 + 4 * 128, 128) and torch.equal(a, ops.mx_block_rearrange_2d_M_groups(s, offs))
    e = torch.ops.torchao.mx_block_rearrange_2d_M_groups(s.view(torch.float8_e8m0fnu), offs, 16)
    assert e.dtype == torch.float8_e8m0fnu and torch.equal(e.view(torch.uint8), a)
    with pytest.raises(RuntimeError, match="chunks_per_tb"):
        torch.ops.torchao.mx_block_rearrange_2d_M_groups(s, offs, 3)
    with pytest.raises(RuntimeError, match="int32"):
        torch.ops.torchao.mx_block_rearrange_2d_M_groups(s, offs.long(), 4)
    with pytest.raises(RuntimeError, match="uint8 or e8m0"):
        torch.ops.torchao.mx_block_rearrange_2d_M_groups(s.to(torch.int8), offs, 4)


def test_cxx_aten_signature_ops_match_the_c_abi_wrappers():
    """ao_mi355_c::* carry the ATen schemas (what the opt-in override installs under aten::) and run the same kernels."""
    n, k, g = 256, 1024, 128
    w = _randn_bf16((n, k), 5, 0.05).to(DEV)
    x = _randn_bf16((3, k), 6).to(DEV)
    qdata, sz = ops.int4_quantize_tinygemm(w, g)
    assert torch.equal(torch.ops.ao_mi355_c._weight_int4pack_mm(x, qdata, g, sz), ops.weight_int4pack_mm(x, qdata, g, sz))
    assert torch.equal(torch.ops.ao_mi355_c._convert_weight_to_int4pack(ops.unpack_int4pack(qdata), 8), qdata)
    wq, ws = ops.int8_quantize_rowwise(w)
    xq, xs = ops.int8_quantize_rowwise(x)
    assert torch.equal(torch.ops.ao_mi355_c._int_mm(xq, wq.t()), ops.int_mm(xq, wq.t()))
    fq, fs = ops.fp8_quantize_rowwise(w)
    aq, a_s = ops.fp8_quantize_rowwise(x)
    bias = _randn_bf16((n,), 7).to(DEV)
    y = torch.ops.ao_mi355_c._scaled_mm(aq, fq.t(), a_s, fs.t(), bias, None, torch.bfloat16, True)
    assert torch.equal(y, ops.fp8_scaled_mm(aq, fq.t(), a_s, fs.t(), bias))
    # tensorwise scales ([1, 1] both: the reference's default Float8DynamicActivationFloat8WeightConfig) broadcast in the binding
    tq, ts = ops.fp8_quantize_tensorwise(w)
    uq, us = ops.fp8_quantize_tensorwise(x)
    y = torch.ops.ao_mi355_c._scaled_mm(uq, tq.t(), us, ts, bias, None, torch.bfloat16, True)
    assert torch.equal(y, ops.fp8_scaled_mm(uq, tq.t(), us.reshape(-1).expand(3), ts.reshape(-1).expand(n), bias))
    with pytest.raises(RuntimeError, match="rowwise"):  # anything else (here: 2 scales for 3 rows) is refused
        torch.ops.ao_mi355_c._scaled_mm(aq, fq.t(), a_s[:2], fs.t(), None, None, torch.bfloat16, True)
    # grouped MXFP8: mat2 arrives [E, K, N] with K-major experts (the transpose of [E, N, K])
    E = 4
    we = _randn_bf16((E, 64, 512), 8, 0.1).to(DEV)
    a = _randn_bf16((96, 512), 9).to(DEV)
    offs = torch.tensor([32, 32, 64, 96], dtype=torch.int32, device=DEV)
    wd, wsc = ops.mxfp8_quantize(we, "rceil")
    ad, asc = ops.mxfp8_quantize(a, "rceil")
    y = torch.ops.ao_mi355_c._scaled_grouped_mm(ad, wd.transpose(1, 2), asc, wsc, offs, None, None, torch.bfloat16, False)
    assert torch.equal(y, ops.mxfp8_grouped_mm(ad, asc, wd, wsc, offs))


def test_opcheck_custom_ops():
    """Schema / fake-kernel / dispatch consistency of the custom ops the subclasses trace through."""
    from torch.library import opcheck

    n, k = 64, 256
    w = _randn_bf16((n, k), 11, 0.05).to(DEV)
    x = _randn_bf16((4, k), 12).to(DEV)
    qdata, sz = ops.int4_quantize_tinygemm(w, 128)
    wq, ws = ops.int8_quantize_rowwise(w)
    fq, fs = ops.fp8_quantize_rowwise(w)
    utils = ("test_schema", "test_faketensor")
    fake = ("test_faketensor",)  # opcheck's schema test clones / compares its inputs and outputs with ops float8 tensors lack
    opcheck(torch.ops.ao_mi355.weight_int4pack_mm.default, (x, qdata, 128, sz), test_utils=utils)
    opcheck(torch.ops.ao_mi355.int8_linear.default, (x, wq, ws, None), test_utils=utils)
    opcheck(torch.ops.ao_mi355.fp8_linear.default, (x, fq, fs, None), test_utils=fake)
    opcheck(torch.ops.ao_mi355.int8_quantize_rowwise.default, (x,), test_utils=utils)
    opcheck(torch.ops.ao_mi355.fp8_quantize_rowwise.default, (x,), test_utils=fake)
    x32 = _randn_bf16((64, 256), 13).to(DEV)
    opcheck(torch.ops.torchao.mxfp8_quantize.default, (x32, True, True, 32, 32, "e4m3", "rceil"), test_utils=fake)
    offs = torch.tensor([10, 64], dtype=torch.int32, device=DEV)
    opcheck(torch.ops.torchao.fused_pad_token_groups.default, (x32, offs, 32), test_utils=utils)
    sc = torch.randint(0, 256, (64, 8), dtype=torch.uint8, device=DEV)
    opcheck(torch.ops.torchao.mx_block_rearrange_2d_M_groups.default, (sc, offs, 4), test_utils=utils)


@pytest.mark.parametrize("kind", ["int4", "int4_plain", "int8", "fp8", "int8_asym", "int8_pt", "fp8_pt", "int8_static", "fp8_clamped"])
def test_torch_compile_fullgraph_through_the_subclass(kind):
    """torch.compile(fullgraph=True) traces F.linear on the quantized weight to ONE extern call of the ao_mi355:: op (the
    reference asserts the same shape of graph: extern_kernels._int_mm, test_int8_tensor.py:276-278) and reproduces eager."""
    from torch._dynamo.utils import counters

    from ao_amd.quantization import (Float8DynamicActivationFloat8WeightConfig, Int4WeightOnlyConfig,
                                     Int8DynamicActivationInt8WeightConfig, Int8StaticActivationInt8WeightConfig, MappingType, PerRow,
                                     PerTensor, quantize_)

    torch.manual_seed(0)
    lin = torch.nn.Linear(1024, 256, bias=True).to(torch.bfloat16).to(DEV)
    cfg = {"int4": Int4WeightOnlyConfig(group_size=128, int4_packing_format="tile_packed_to_4d"),
           "int4_plain": Int4WeightOnlyConfig(group_size=128),  # the DEFAULT packing format (PLAIN): its compute layout is an inner tensor
           "int8": Int8DynamicActivationInt8WeightConfig(),
           "fp8": Float8DynamicActivationFloat8WeightConfig(granularity=PerRow()),
           "int8_asym": Int8DynamicActivationInt8WeightConfig(act_mapping_type=MappingType.ASYMMETRIC),
           "int8_pt": Int8DynamicActivationInt8WeightConfig(granularity=PerTensor()),
           "fp8_pt": Float8DynamicActivationFloat8WeightConfig(),
           "int8_static": Int8StaticActivationInt8WeightConfig(act_quant_scale=torch.tensor([[0.05]], device=DEV), granularity=[PerTensor(), PerRow()]),
           "fp8_clamped": Float8DynamicActivationFloat8WeightConfig(granularity=PerRow(), activation_value_lb=0.5, activation_value_ub=3.0)}[kind]
    quantize_(lin, cfg)
    x = _randn_bf16((5, 1024), 21).to(DEV)
    want = lin(x)
    torch._dynamo.reset()
    counters.clear()
    seen = []

    def fw_compiler(gm, example_inputs):  # the graph after AOT autograd has desugared the subclass (what inductor receives)
        seen.extend(str(n.target) for n in gm.graph.nodes if n.op == "call_function")
        return gm.forward

    from torch._dynamo.backends.common import aot_autograd

    got = torch.compile(lin, fullgraph=True, backend=aot_autograd(fw_compiler=fw_compiler))(x)
    assert torch.equal(got, want)
    opname = {"int4": "ao_mi355.weight_int4pack_mm", "int4_plain": "ao_mi355.weight_int4pack_mm", "int8": "ao_mi355.int8_linear.", "fp8": "ao_mi355.fp8_linear.",
              "int8_asym": "ao_mi355.int8_linear_asym", "int8_pt": "ao_mi355.int8_linear_tensorwise", "fp8_pt": "ao_mi355.fp8_linear_tensorwise",
              "int8_static": "ao_mi355.int8_linear_static", "fp8_clamped": "ao_mi355.fp8_linear_clamped"}[kind]
    assert sum(opname in t for t in seen) == 1, seen
    assert counters["graph_break"] == {} or sum(counters["graph_break"].values()) == 0
    # and through inductor (the op becomes an extern kernel call)
    torch._dynamo.reset()
    got2 = torch.compile(lin, fullgraph=True)(x)
    assert torch.equal(got2, want)


_OVERRIDE_SCRIPT = r"""
import ctypes, sys, torch
sys.path.insert(0, {root!r})
from ao_amd import _lib, ops, torch_ops
assert torch_ops.aten_overrides_installed(), "AO_MI355_OVERRIDE_ATEN=1 did not activate the C++ aten overrides"
dev = "cuda"
g = torch.Generator().manual_seed(0)
w = (torch.randn(256, 1024, generator=g) * 0.05).to(torch.bfloat16).to(dev)
x = torch.randn(3, 1024, generator=g).to(torch.bfloat16).to(dev)
qdata, sz = ops.int4_quantize_tinygemm(w, 128)
wq, ws = ops.int8_quantize_rowwise(w); xq, xs = ops.int8_quantize_rowwise(x)
fq, fs = ops.fp8_quantize_rowwise(w); aq, a_s = ops.fp8_quantize_rowwise(x)
want = (ops.weight_int4pack_mm(x, qdata, 128, sz), ops.int_mm(xq, wq.t()), ops.fp8_scaled_mm(aq, fq.t(), a_s, fs.t()))
lib = _lib.lib()
_lib.check(lib.ao_prof_enable(8))
got = (torch.ops.aten._weight_int4pack_mm(x, qdata, 128, sz), torch._int_mm(xq, wq.t()),
       torch._scaled_mm(aq, fq.t(), scale_a=a_s, scale_b=fs.t(), out_dtype=torch.bfloat16, use_fast_accum=True))
q2 = torch.ops.aten._convert_weight_to_int4pack(ops.unpack_int4pack(qdata), 8)
buf, cnt = (ctypes.c_float * 8)(), ctypes.c_int(0)
_lib.check(lib.ao_prof_collect(buf, 8, ctypes.byref(cnt)))
assert cnt.value == 5, cnt.value  # int4 mm, int mm, scaled mm, unpack, pack: all launched by ao_amd/_C_mi355.so
assert all(torch.equal(a, b) for a, b in zip(got, want)) and torch.equal(q2, qdata)
print("override ok")
"""


def test_aten_overrides_opt_in_fresh_process():
    env = dict(os.environ, AO_MI355_OVERRIDE_ATEN="1")
    out = subprocess.run([sys.executable, "-c", _OVERRIDE_SCRIPT.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "override ok" in out.stdout, out.stdout + out.stderr


def test_torch_compile_fp16_activation_refuses_with_the_reason():
    """fp16 / fp32 activations take an eager-only slow path (raw int32 GEMM, no fake kernel): under torch.compile the linear refuses with the
    message that names the dtype, not with a FakeTensor data_ptr error; eager still works."""
    from ao_amd.quantization import Int8DynamicActivationInt8WeightConfig, quantize_

    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 128, bias=False).to(torch.bfloat16).to(DEV)
    quantize_(lin, Int8DynamicActivationInt8WeightConfig())
    w = lin.weight
    x = torch.randn(4, 512, device=DEV, dtype=torch.float16)
    assert torch.nn.functional.linear(x, w).dtype == torch.float16  # eager: the slow path
    torch._dynamo.reset()
    with pytest.raises(Exception, match="takes bfloat16 activations"):
        torch.compile(lambda a: torch.nn.functional.linear(a, w), fullgraph=True, backend="aot_eager")(x)
    torch._dynamo.reset()
