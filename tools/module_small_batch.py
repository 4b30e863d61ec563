"""module fwd+bwd (projections + core + autograd) at several batch sizes, one autograd node each, NCHW fp32 tensors: NCHW strip
family vs split-plane family (the default route), and the pixel-major family on channels_last tensors"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ccnet_amd import CrissCrossAttention
dev = torch.device("cuda:0")
C, H, W = 512, 97, 97
for B in (1, 2, 4, 8):
    row, ys = [], []
    for planes, cl in ((False, False), (True, False), (False, True)):
        torch.manual_seed(0)
        m = CrissCrossAttention(C).to(dev)
        m.split_planes = planes
        with torch.no_grad():
            m.gamma.fill_(0.5)
        x = torch.randn(B, C, H, W, device=dev)
        dy = torch.randn(B, C, H, W, device=dev)
        if cl:
            x, dy = x.contiguous(memory_format=torch.channels_last), dy.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        def one():
            m.zero_grad(set_to_none=True); x.grad = None
            y = m(x); y.backward(dy); return y
        for _ in range(3): y = one()
        torch.cuda.synchronize()
        ys.append((y.detach().clone(), x.grad.clone(), m.value_conv.weight.grad.clone()))
        row.append(bench.time_region(one, 20))
    d = [float((a - b).abs().max()) for a, b in zip(ys[0], ys[1])]
    print(f"B={B}: module fwd+bwd  NCHW-strip node {row[0]:.3f} ms | split-plane node (default) {row[1]:.3f} ms | pixel-major "
          f"family, channels_last tensors {row[2]:.3f} ms   (max |diff| strip vs split-plane: y {d[0]:.1e} dx {d[1]:.1e} "
          f"dWv {d[2]:.1e})", flush=True)
