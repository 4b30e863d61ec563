#!/usr/bin/env python3
"""A/B of the module's forward forms (CrissCrossAttention.projection_forward_mode, round 5) at (8,512,97,97) fp32: one stacked GEMM +
fused forward (0), two GEMMs on one stream (1), two GEMMs with the affinity + softmax launches next to the v GEMM (2).  Per mode:
module forward (no_grad), forward with autograd, fwd+bwd; two rounds; outputs bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ccnet_amd import CrissCrossAttention  # noqa: E402

B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 512, 97, 97)
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = CrissCrossAttention(C).to(dev)
with torch.no_grad():
    m.gamma.fill_(0.5)
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
dy = torch.randn(B, C, H, W, device=dev)


def step():
    m.zero_grad(set_to_none=True)
    x.grad = None
    m(x).backward(dy)


def infer():
    with torch.no_grad():
        return m(x)


ref = None
for rnd in range(2):
    for mode in (2, 1, 0):
        m.projection_forward_mode = mode
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t_step = bench.time_region(step, 20)
        t_inf = bench.time_region(infer, 20)
        t_fwd = bench.time_region(lambda: m(x), 20)
        y = infer()
        step()
        g = (y.clone(), x.grad.clone(), m.value_conv.weight.grad.clone())
        if ref is None:
            ref = g
        same = all(torch.equal(a, b) for a, b in zip(g, ref))
        print(f"== round {rnd} projection_forward_mode={mode}: fwd+bwd {t_step:.4f} ms  forward (autograd) {t_fwd:.4f}  forward (no_grad) {t_inf:.4f}  bit-identical to first: {same}", flush=True)
