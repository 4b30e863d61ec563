"""module fwd+bwd (projections + core + autograd) at small batches: NCHW strip route vs pixel-major route, NCHW and channels_last inputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ccnet_amd import CrissCrossAttention
dev = torch.device("cuda:0")
C, H, W = 512, 97, 97
for B in (1, 2, 3, 4, 8):
    row = []
    for pm, cl in ((0, False), (B, False), (B, True)):
        m = CrissCrossAttention(C).to(dev)
        m.small_batch_pixel_major = pm
        with torch.no_grad():
            m.gamma.fill_(0.5)
        x = torch.randn(B, C, H, W, device=dev)
        dy = torch.randn(B, C, H, W, device=dev)
        if cl:
            x, dy = x.contiguous(memory_format=torch.channels_last), dy.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        def one():
            m.zero_grad(set_to_none=True); x.grad = None
            m(x).backward(dy)
        for _ in range(3): one()
        torch.cuda.synchronize()
        row.append(bench.time_region(one, 20))
    print(f"B={B}: module fwd+bwd  NCHW-strip route {row[0]:.3f} ms | pixel-major route, NCHW x {row[1]:.3f} ms | pixel-major route, channels_last x {row[2]:.3f} ms", flush=True)
