#!/usr/bin/env python3
"""The default module fwd+bwd at (8,512,97,97), a few steps, for rocprofv3 --kernel-trace --stats (every launch of the step, the
library's and torch's)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccnet_amd import CrissCrossAttention  # noqa: E402

dev = torch.device("cuda:0")
B, C, H, W = (int(a) for a in sys.argv[2:6]) if len(sys.argv) >= 6 else (8, 512, 97, 97)
torch.manual_seed(0)
m = CrissCrossAttention(C).to(dev)
with torch.no_grad():
    m.gamma.fill_(0.5)
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
dy = torch.randn(B, C, H, W, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    m.zero_grad(set_to_none=True)
    x.grad = None
    m(x).backward(dy)
torch.cuda.synchronize()
print("ok")
