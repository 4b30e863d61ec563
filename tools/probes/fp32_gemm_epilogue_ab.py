#!/usr/bin/env python3
"""The fp32 projection GEMMs of the module at small batches (B = 1, 2, 4: below split_bf16_min_pixels): stock epilogues (bias / beta = 1)
against the bare product + an elementwise pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
C, H, W = 512, 97, 97
hw, ct = H * W, C + 2 * (C // 8)
for B in (1, 2, 4):
    torch.manual_seed(0)
    x = torch.randn(B, C, hw, device=dev)
    w = torch.randn(ct, C, device=dev)
    b = torch.randn(ct, device=dev)
    dy = torch.randn(B, C, hw, device=dev)
    dq = torch.randn(B, hw, ct, device=dev)
    xt, wt, wtt, dqt = x.transpose(1, 2), w.t().unsqueeze(0).expand(B, -1, -1), w.t().unsqueeze(0).expand(B, -1, -1), dq.transpose(1, 2)
    fw = {"fwd baddbmm(bias, x^T, W^T)  [ships]": lambda: torch.baddbmm(b.view(1, 1, -1), xt, wt),
          "fwd bmm + add_(bias)": lambda: torch.bmm(xt, wt).add_(b),
          "bwd baddbmm(dy, W^T, dqkv^T)  [ships]": lambda: torch.baddbmm(dy, wtt, dqt),
          "bwd bmm + add_(dy)": lambda: torch.bmm(wtt, dqt).add_(dy)}
    for rnd in range(2):
        for name, f in fw.items():
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            print(f"B={B} round {rnd}: {name:40s} {bench.time_region(f, 30) * 1e3:8.1f} us", flush=True)
