#!/usr/bin/env python3
"""Time the fused core fwd+bwd at an arbitrary shape / dtype (which kernel family serves it is printed).
usage: python tools/time_shape.py B C H W [f32|bf16] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ccnet_amd import _lib
from ccnet_amd.functions import CrissCrossBF16Function, CrissCrossFunction

B, C, H, W = (int(a) for a in sys.argv[1:5])
dt = torch.bfloat16 if len(sys.argv) > 5 and sys.argv[5] == "bf16" else torch.float32
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
lib = _lib.get_lib()
torch.manual_seed(0)
q, k = (torch.randn(B, C // 8, H, W, device=dev).mul_(0.3).to(dt).requires_grad_(True) for _ in range(2))
v, x = (torch.randn(B, C, H, W, device=dev).to(dt).requires_grad_(True) for _ in range(2))
g = torch.full((1,), 0.5, device=dev, requires_grad=True)
dy = torch.randn(B, C, H, W, device=dev).to(dt)
fn = CrissCrossBF16Function if dt == torch.bfloat16 else CrissCrossFunction

def step():
    for t in (q, k, v, x, g):
        t.grad = None
    fn.apply(q, k, v, x, g).backward(dy)

step(); torch.cuda.synchronize()
ms = bench.time_region(step, iters)
nbytes = bench.core_bytes(B, C, H, W) * (2 if dt == torch.bfloat16 else 4) // 4
print(f"({B},{C},{H},{W}) {dt}: {ms:.3f} ms per fwd+bwd, {nbytes / ms / 1e6:.1f} GB/s algorithmic, "
      f"mfma strip kernels: {bool(lib.ccnet_cca_shape_uses_mfma(B, C, H, W))}")
if len(sys.argv) > 7 and sys.argv[7] == "stock" and dt == torch.float32:
    print("stock PyTorch formulation:", bench.stock_pytorch_core(B, C, H, W, dev, iters=3))
