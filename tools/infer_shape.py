#!/usr/bin/env python3
"""INFERENCE timing of the attention module (torch.no_grad(), forward only: what evaluate.py:246 runs) at a shape: the module on
its default route, the module forced onto the NCHW strip / windowed kernels, and the reference's bmm / cat / softmax formulation
of the same module (three convolutions + functions.py:30-49) with torch ops; the three outputs are compared.
usage: python tools/infer_shape.py B C H W [iters]      (the reference's whole-image evaluation: 1 512 129 257)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ccnet_amd import CrissCrossAttention

B, C, H, W = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = CrissCrossAttention(C).to(dev).eval()
with torch.no_grad():
    m.gamma.fill_(0.5)
x = torch.randn(B, C, H, W, device=dev)


def stock(x):
    return bench.reference_formulation(m.query_conv(x), m.key_conv(x), m.value_conv(x), x, m.gamma)


with torch.no_grad():
    route = m.route(x)
    y = m(x)
    m.split_planes = False
    route_s = m.route(x)
    ys = m(x)
    yr = stock(x)
    torch.cuda.synchronize()
    t_s = bench.time_region(lambda: m(x), iters)
    m.split_planes = True
    t = bench.time_region(lambda: m(x), iters)
    t_r = bench.time_region(lambda: stock(x), max(3, iters // 4))
print(f"({B},{C},{H},{W}) fp32 inference (module forward, no_grad): route '{route}' {t:.3f} ms | route '{route_s}' {t_s:.3f} ms | "
      f"stock formulation {t_r:.3f} ms  ({t_r / t:.1f}x / {t_r / t_s:.1f}x)   max |y - stock| {float((y - yr).abs().max()):.1e}, "
      f"max |y(strips) - stock| {float((ys - yr).abs().max()):.1e}")
