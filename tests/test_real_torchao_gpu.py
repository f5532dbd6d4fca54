"""End to end with the REAL torchao: an unmodified `torchao.quantize_` + the reference's own tensor
subclasses reach the MI355X kernels through the aten:: overrides of `_C_mi355_ops.so`, and produce the bits the mirror produces.

Needs a GPU AND an importable torchao.  The image of the GPU pool ships none, so `make -C oracle ref` (run by
`__graft_entry__.build()` whenever the read-only reference checkout is present) stages the reference's Python package into the
git-ignored `oracle/_ref/`, which travels to the GPU box with the repository snapshot; `TORCHAO_PATH=/path/to/ao` overrides it, an
installed torchao is the last resort.  Without any of the three the tests skip.
Each case runs in a fresh process: AO_MI355_OVERRIDE_ATEN=1 has to be in the environment when `_C_mi355_ops.so` is loaded.
"""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_STAGED = os.path.join(ROOT, "oracle", "_ref")
TORCHAO_PATH = os.environ.get("TORCHAO_PATH", "") or (_STAGED if os.path.isdir(os.path.join(_STAGED, "torchao")) else "")


def _torchao_importable():
    if TORCHAO_PATH and os.path.isdir(os.path.join(TORCHAO_PATH, "torchao")):
        return True
    return importlib.util.find_spec("torchao") is not None


_SCRIPT = r"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, {root!r})
if {tpath!r}:
    sys.path.insert(0, {tpath!r})
os.environ.setdefault("TORCHAO_FORCE_SKIP_LOADING_SO_FILES", "1")  # the checkout's CUDA .so files are not for this box
import torchao
from torchao.quantization import quantize_ as ref_quantize
from ao_amd import _lib, torch_ops
import ao_amd.quantization as mine
assert torch_ops.aten_overrides_installed(), "AO_MI355_OVERRIDE_ATEN=1 did not activate the aten overrides"
lib = _lib.lib()
dev = "cuda"
kind = {kind!r}
torch.manual_seed(0)
n, k = 512, 2048
lin_ref = torch.nn.Linear(k, n, bias=True).to(torch.bfloat16).to(dev)
lin_mine = torch.nn.Linear(k, n, bias=True).to(torch.bfloat16).to(dev)
lin_mine.load_state_dict(lin_ref.state_dict())
x = torch.randn(7, k, dtype=torch.bfloat16, device=dev)
if kind == "int4":
    from torchao.quantization import Int4WeightOnlyConfig
    ref_quantize(lin_ref, Int4WeightOnlyConfig(group_size=128, int4_packing_format="tile_packed_to_4d"))
    mine.quantize_(lin_mine, mine.Int4WeightOnlyConfig(group_size=128, int4_packing_format="tile_packed_to_4d"))
    same_weight = bool(torch.equal(lin_ref.weight.qdata, lin_mine.weight.qdata) and torch.equal(lin_ref.weight.scale_and_zero, lin_mine.weight.scale_and_zero))
elif kind == "int8":
    from torchao.quantization import Int8DynamicActivationInt8WeightConfig
    ref_quantize(lin_ref, Int8DynamicActivationInt8WeightConfig(version=2) if "version" in Int8DynamicActivationInt8WeightConfig.__dataclass_fields__ else Int8DynamicActivationInt8WeightConfig())
    mine.quantize_(lin_mine, mine.Int8DynamicActivationInt8WeightConfig())
    same_weight = bool(torch.equal(lin_ref.weight.qdata, lin_mine.weight.qdata) and torch.equal(lin_ref.weight.scale.reshape(-1), lin_mine.weight.scale.reshape(-1)))
else:
    from torchao.quantization import Float8DynamicActivationFloat8WeightConfig, PerRow
    ref_quantize(lin_ref, Float8DynamicActivationFloat8WeightConfig(granularity=PerRow()))
    mine.quantize_(lin_mine, mine.Float8DynamicActivationFloat8WeightConfig(granularity=mine.PerRow()))
    same_weight = bool(torch.equal(lin_ref.weight.qdata.view(torch.uint8), lin_mine.weight.qdata.view(torch.uint8))
                       and torch.equal(lin_ref.weight.scale.reshape(-1), lin_mine.weight.scale.reshape(-1)))
assert type(lin_ref.weight).__module__.startswith("torchao."), type(lin_ref.weight)
# launches of OUR library while the reference's F.linear runs (ao_prof counts only launches made through _C_mi355.so)
assert lib.ao_prof_enable(64) == 0
y_ref = lin_ref(x)
buf = (ctypes.c_float * 64)(); cnt = ctypes.c_int(0)
assert lib.ao_prof_collect(buf, 64, ctypes.byref(cnt)) == 0
y_mine = lin_mine(x)
torch.cuda.synchronize()
rel = float((y_ref.float() - y_mine.float()).norm() / y_mine.float().norm())
print(json.dumps({{"kind": kind, "weight_class": type(lin_ref.weight).__name__, "same_weight_bits": same_weight, "our_launches_inside_reference_linear": cnt.value,
                  "output_equal": bool(torch.equal(y_ref, y_mine)), "rel": rel, "torchao": getattr(torchao, "__version__", "?")}}))
"""


@pytest.mark.skipif(not _torchao_importable(), reason="the real torchao is not importable here (set TORCHAO_PATH to a checkout)")
@pytest.mark.parametrize("kind", ["int4", "int8", "fp8"])
def test_unmodified_torchao_reaches_the_kernels(kind):
    env = dict(os.environ, AO_MI355_OVERRIDE_ATEN="1")
    out = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT, tpath=TORCHAO_PATH, kind=kind)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    log = os.path.join(ROOT, "gpurun_out", "real_torchao.jsonl")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "a") as f:
        f.write(json.dumps(rec) + "\n")
    assert rec["our_launches_inside_reference_linear"] >= 1, rec   # the reference's F.linear ran on our kernels
    assert rec["same_weight_bits"], rec                              # reference weight prep == mirror weight prep, bit for bit
    if kind == "fp8":
        assert rec["rel"] <= 1e-3, rec
    else:
        assert rec["output_equal"], rec
