#!/usr/bin/env python3
"""Where the split-bf16 projection path (the library's GEMMs, csrc/cca_gemm.hpp) starts to pay: module fwd+bwd at (B,512,97,97),
B = 1, 2, 3, 4, with ``split_bf16_min_pixels`` at 32768 (rounds 3-5) and at 0 (round 6)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ccnet_amd import CrissCrossAttention  # noqa: E402

dev = torch.device("cuda:0")
default = 32768                    # the default of rounds 3-5 (stock GEMMs); 0 since round 6
shapes = [(1, 512, 97, 97), (2, 512, 97, 97), (3, 512, 97, 97), (4, 512, 97, 97), (8, 512, 97, 97),
          (1, 512, 33, 33), (2, 512, 49, 49), (1, 512, 65, 65), (2, 512, 65, 65), (1, 256, 97, 97), (2, 64, 20, 24), (1, 512, 129, 129)]
for B, C, H, W in shapes:
    torch.manual_seed(0)
    m = CrissCrossAttention(C).to(dev)
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    dy = torch.randn(B, C, H, W, device=dev)

    def one():
        m.zero_grad(set_to_none=True)
        x.grad = None
        m(x).backward(dy)

    row = []
    for thr in (default, 0, default, 0):
        m.split_bf16_min_pixels = thr
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        row.append(bench.time_region(one, 20))
    print(f"({B},{C},{H},{W}) ({B * H * W} pixels): threshold {default}: {row[0]:.3f} / {row[2]:.3f} ms   threshold 0 (split-bf16 GEMMs): {row[1]:.3f} / {row[3]:.3f} ms", flush=True)
