#!/usr/bin/env python3
"""Round 5: the module's backward needs dqkv as three-plane rows AND its column sums.  Times, at (B,97,97,640) fp32: split_planes alone,
torch's sum over pixels alone, and ccnet_cca_split_planes_colsum_f32 (both in one pass) -- with the launch profiler's per-launch split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ccnet_amd import _lib  # noqa: E402
from ccnet_amd.functions import PLANES_HLH, split_planes, split_planes_colsum  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
for B in (8, 4, 1):
    t = torch.randn(B, 97, 97, 640, device=dev)
    for _ in range(3):
        split_planes(t, 0, 640, PLANES_HLH, torch.bfloat16); t.sum(dim=(0, 1, 2)); split_planes_colsum(t)
    torch.cuda.synchronize()
    a = bench.time_region(lambda: split_planes(t, 0, 640, PLANES_HLH, torch.bfloat16), 20) * 1e3
    b = bench.time_region(lambda: t.view(B, -1, 640).sum(dim=(0, 1)), 20) * 1e3
    c = bench.time_region(lambda: split_planes_colsum(t), 20) * 1e3
    rec = lib.profile_launches(lambda: split_planes_colsum(t))
    d3, cs = split_planes_colsum(t)
    ok = torch.equal(d3, split_planes(t, 0, 640, PLANES_HLH, torch.bfloat16))
    err = float((cs - t.double().sum(dim=(0, 1, 2)).float()).abs().max())
    print(f"B={B}: split_planes {a:.1f} us + torch sum {b:.1f} us  vs  one pass {c:.1f} us  {[(n.replace('cca::', '')[:28], round(ms * 1e3, 1)) for n, ms in rec]}  planes identical {ok}, max |colsum - fp64 sum| {err:.2e}")
