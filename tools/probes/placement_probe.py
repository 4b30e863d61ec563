#!/usr/bin/env python3
"""Probe: how much do the forward / backward times of the SAME kernels on the SAME shape move with where the caching allocator
happens to place the tensors?  Builds several PlanesWorkload instances (earlier ones stay alive, small pads shift the next
one's addresses) and prints time + the addresses modulo a few powers of two."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device("cuda:0")
B, C, H, W = 8, 512, 97, 97
keep = []
for i in range(8):
    if i:
        keep.append(torch.empty((i * 1237 * 4096 + 256 * i) // 4, device=dev))   # shift what the allocator hands out next
    wl = bench.PlanesWorkload(lib, B, C, H, W, dev, 1234)
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    f = min(bench.time_region(wl.forward, 20) for _ in range(3))
    b = min(bench.time_region(wl.backward, 20) for _ in range(3))
    rec = lib.profile_launches(lambda: [wl.forward() for _ in range(5)])
    n = len(rec) // 5
    per = [sum(rec[r * n + k][1] for r in range(5)) / 5 * 1e3 for k in range(n)]
    ptrs = {nm: getattr(wl, nm).data_ptr() for nm in ("qkv", "vpl", "x", "y", "A", "ws")}
    print(f"instance {i}: fwd {f:.4f} bwd {b:.4f} ms  fwd launches us {[round(p, 1) for p in per]}  "
          + " ".join(f"{nm}%2M={p % (1 << 21) >> 12:4d}p" for nm, p in ptrs.items()), flush=True)
    keep.append(wl)
