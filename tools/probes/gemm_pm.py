#!/usr/bin/env python3
"""The three projection GEMMs of the module at (8,512,97,97), a few launches each, for counter passes
(bash tools/pmc.sh <tag> --script tools/probes/gemm_pm.py): csrc/cca_gemm.hpp."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccnet_amd import functions as F, _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.get_lib()
B, C, H, W = 8, 512, 97, 97
hw, ct = H * W, C + 2 * (C // 8)
torch.manual_seed(0)
x3 = torch.randn(B * hw, 3 * C, device=dev).to(torch.bfloat16)
w3 = torch.randn(ct, 3 * C, device=dev).to(torch.bfloat16)
w3t = torch.randn(C, 3 * ct, device=dev).to(torch.bfloat16)
d3 = torch.randn(B, hw, 3 * ct, device=dev).to(torch.bfloat16)
dy = torch.randn(B, C, hw, device=dev)
bias = torch.randn(ct, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    F._projection_gemm(lib, x3, w3, bias)
    F._projection_adjoint_gemm(lib, w3t, d3, dy)
    F._projection_wgrad_gemm(lib, d3.view(B * hw * 3, ct), x3.view(B * hw * 3, C))
torch.cuda.synchronize()
print("ok")
