#!/usr/bin/env python3
"""Time the attention at an arbitrary shape / dtype THROUGH THE AUTOGRAD PATH a user runs: the whole module (projections + core +
backward; the route it takes is printed) and, for fp32, the fused core alone through ``CrissCrossFunction`` (NCHW strip family).
usage: python tools/time_shape.py B C H W [f32|bf16] [iters] [stock]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ccnet_amd import CrissCrossAttention, _lib
from ccnet_amd.functions import CrissCrossFunction

B, C, H, W = (int(a) for a in sys.argv[1:5])
dt = torch.bfloat16 if len(sys.argv) > 5 and sys.argv[5] == "bf16" else torch.float32
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
lib = _lib.get_lib()
torch.manual_seed(0)
m = CrissCrossAttention(C).to(dev).to(dt)
with torch.no_grad():
    m.gamma.fill_(0.5)
xm = torch.randn(B, C, H, W, device=dev).to(dt).requires_grad_(True)
dy = torch.randn(B, C, H, W, device=dev).to(dt)

def mod_step():
    m.zero_grad(set_to_none=True)
    xm.grad = None
    m(xm).backward(dy)

for _ in range(2):
    mod_step()
torch.cuda.synchronize()
msm = bench.time_region(mod_step, iters)
nbytes = bench.core_bytes(B, C, H, W) * (2 if dt == torch.bfloat16 else 4) // 4
print(f"({B},{C},{H},{W}) {dt}: module fwd+bwd {msm:.3f} ms on route '{m.route(xm)}'")
if dt == torch.float32:
    q, k = (torch.randn(B, C // 8, H, W, device=dev).mul_(0.3).requires_grad_(True) for _ in range(2))
    v, x = (torch.randn(B, C, H, W, device=dev).requires_grad_(True) for _ in range(2))
    g = torch.full((1,), 0.5, device=dev, requires_grad=True)

    def step():
        for t in (q, k, v, x, g):
            t.grad = None
        CrissCrossFunction.apply(q, k, v, x, g).backward(dy)

    step(); torch.cuda.synchronize()
    ms = bench.time_region(step, iters)
    print(f"    core alone through CrissCrossFunction (NCHW strip family): {ms:.3f} ms, {nbytes / ms / 1e6:.1f} GB/s algorithmic, "
          f"kernel family {lib.ccnet_cca_shape_uses_mfma(B, C, H, W)} (1 stationary strips, 2 windowed, 0 any-shape)")
    if len(sys.argv) > 7 and sys.argv[7] == "stock":
        print("    stock PyTorch formulation of the core:", bench.stock_pytorch_core(B, C, H, W, dev, iters=3))
