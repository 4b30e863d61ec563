"""Shared fixtures.  GPU tests are marked ``@pytest.mark.gpu`` and call through the C-ABI library."""
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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """Load a fixture written by tests/golden/make_golden.py as a dict of torch tensors."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def regenerate_module_inputs(B, C, H, W):
    """Same RNG sequence as make_golden.run_case: Conv2d default init, then x, then dy."""
    torch.manual_seed(0)
    q = torch.nn.Conv2d(C, C // 8, 1)
    k = torch.nn.Conv2d(C, C // 8, 1)
    v = torch.nn.Conv2d(C, C, 1)
    params = {
        "gamma": torch.full((1,), 0.5),
        "query_conv.weight": q.weight.detach(), "query_conv.bias": q.bias.detach(),
        "key_conv.weight": k.weight.detach(), "key_conv.bias": k.bias.detach(),
        "value_conv.weight": v.weight.detach(), "value_conv.bias": v.bias.detach(),
    }
    x = torch.randn(B, C, H, W)
    dy = torch.randn(B, C, H, W)
    return x, dy, params


def make_core_inputs(B, C, H, W, seed=0, dtype=torch.float32, device="cpu", scale=1.0):
    """Seeded q, k, v, x, dy for core-level (post-conv) parity tests."""
    g = torch.Generator().manual_seed(seed)
    Cq = max(C // 8, 1)
    q = torch.randn(B, Cq, H, W, generator=g) * scale
    k = torch.randn(B, Cq, H, W, generator=g) * scale
    v = torch.randn(B, C, H, W, generator=g)
    x = torch.randn(B, C, H, W, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    return [t.to(dtype=dtype, device=device) for t in (q, k, v, x, dy)]


SMALL_CASES = ["tiny_2x16x5x6", "small_1x32x9x7", "small_2x64x8x8"]
