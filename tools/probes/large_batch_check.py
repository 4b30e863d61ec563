#!/usr/bin/env python3
"""Module fwd+bwd at large batches with the library's projection GEMMs against the same node on the stock GEMMs (the fallback
pair of every helper): offsets near the 31-bit limits of the entry points, and the batch from which the helpers hand over."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccnet_amd import CrissCrossAttention, functions as F  # noqa: E402

dev = torch.device("cuda:0")
C, H, W = 512, 97, 97
for B in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (16, 24, 40, 48):
    torch.manual_seed(B)
    m = CrissCrossAttention(C).to(dev)
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    dy = torch.randn(B, C, H, W, device=dev)
    used = {}
    keep = (F._projection_gemm, F._projection_adjoint_gemm, F._projection_wgrad_gemm)

    def spy(name, f):
        def g(*a, **k):
            r = f(*a, **k)
            used[name] = r is not None
            return r
        return g

    res = []
    for lib_gemms in (True, False):
        if lib_gemms:
            F._projection_gemm, F._projection_adjoint_gemm, F._projection_wgrad_gemm = (spy(n, f) for n, f in zip(("fwd", "dx", "dW"), keep))
        else:
            F._projection_gemm = F._projection_adjoint_gemm = F._projection_wgrad_gemm = lambda *a, **k: None
        m.zero_grad(set_to_none=True)
        x.grad = None
        y = m(x)
        y.backward(dy)
        torch.cuda.synchronize()
        res.append((y.detach().clone(), x.grad.clone(), m.value_conv.weight.grad.clone(), m.query_conv.weight.grad.clone()))
    F._projection_gemm, F._projection_adjoint_gemm, F._projection_wgrad_gemm = keep
    d = [float((a - b).abs().max()) / max(1e-30, float(b.abs().max())) for a, b in zip(*res)]
    print(f"B={B}: library GEMMs used {used}; max relative |diff| vs the stock GEMMs: y {d[0]:.1e} dx {d[1]:.1e} dWv {d[2]:.1e} dWq {d[3]:.1e}", flush=True)
    assert all(v < 1e-4 for v in d), d
    del res, x, dy, y
    torch.cuda.empty_cache()
print("ok")
