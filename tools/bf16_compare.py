#!/usr/bin/env python3
"""A/B of BASELINE configs[4] (16,512,129,129) bf16 on the pixel-major family: gmap_kernel / gweight_kernel everywhere
("planes_ring" 0) vs the ring kernel in the column passes (the default; the dA contraction stays on gweight_kernel).
The last variants add the dv passes on the side stream ("planes_overlap" 1 / 2).  Per variant the step time and the in-step duration of every launch; outputs must be bit-identical."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device("cuda:0")
shape = tuple(int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 512, 129, 129)
ref = None
for opts in ((0, 0, 0), (2, 1, 0), (2, 1, 1), (2, 1, 2)):
    lib.set_option("planes_ring", opts[0]); lib.set_option("planes_stream", opts[1])
    lib.set_option("planes_overlap", opts[2])
    wl = bench.PixelMajorBF16Workload(lib, *shape, dev, 1)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    ms = bench.time_region(wl.step, 20)
    rec = lib.profile_launches(lambda: [wl.step() for _ in range(3)])
    n = len(rec) // 3
    print(f"bf16 {shape} planes_ring={opts[0]} planes_stream={opts[1]} planes_overlap={opts[2]}: step {ms:.4f} ms")
    for i in range(n):
        print("    %8.1f us  %s" % (sum(rec[r * n + i][1] for r in range(3)) / 3 * 1e3, rec[i][0][:100]))
    cur = (wl.y.clone(), wl.dqkv.clone())
    if ref is None:
        ref = cur
    else:
        print("    bit-identical to the first variant:", torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]))
lib.set_option("planes_ring", 2); lib.set_option("planes_stream", 1); lib.set_option("planes_overlap", -1)
