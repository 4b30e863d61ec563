import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from ccnet_amd import CrissCrossAttention, _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
C, H, W = 512, 97, 97
for B in (1, 2):
    for ov in (0, 1, 2, -1, 0):
        lib.set_option("planes_overlap", ov)
        torch.manual_seed(0)
        m = CrissCrossAttention(C).to(dev); m.split_bf16_projections = False
        with torch.no_grad(): m.gamma.fill_(0.5)
        x = torch.randn(B, C, H, W, device=dev, requires_grad=True); dy = torch.randn(B, C, H, W, device=dev)
        def one():
            m.zero_grad(set_to_none=True); x.grad = None
            y = m(x); y.backward(dy); return y
        for _ in range(5): one()
        torch.cuda.synchronize()
        print(f"B={B} planes_overlap={ov}: module {bench.time_region(one, 30):.4f} ms", flush=True)
