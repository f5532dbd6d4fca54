import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # the one-process-per-GPU tests (tests/test_multigpu_gpu.py) have never met a multi-GPU node: they run LAST, so that under `-x` a
    # surprise on the first such node cannot hide the results of the single-GPU tests behind it
    items.sort(key=lambda item: "test_multigpu_gpu" in item.nodeid)  # (stable: everything else keeps its order)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def bf16_bits_to_f32(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert np.all((x.view(np.uint32) & 0xFFFF) == 0), "not bf16-representable"
    return (x.view(np.uint32) >> 16).astype(np.uint16)


def torch_bf16_from_f32(x, device="cuda"):
    """bf16-valued fp32 numpy -> torch bf16 tensor (exact)."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(device)


def np_from_torch_bf16(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope="session")
def golden_int4():
    return np.load(os.path.join(GOLDEN, "int4_tinygemm.npz"))
