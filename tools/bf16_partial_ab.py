#!/usr/bin/env python3
"""A/B of BASELINE configs[4] (16,512,129,129) bf16 on the pixel-major family, round 5: the column -> row partial of the
aggregation and of dv as bf16 (option "bf16_partial" 1, the default) against the fp32 partial of rounds 2-4, each with the dv
passes next to softmax-backward / dq | dk ("planes_overlap" 1, the family's default) and next to dA as well (2).  Per variant:
step / fwd / bwd ms and the in-step duration of every launch (single stream); two rounds (the order must not matter); outputs
of the two arithmetic variants differ by one bf16 rounding of the column half (printed: max |y1 - y0| relative to |y|max)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ccnet_amd import _lib  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
shape = tuple(int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 512, 129, 129)
wl = bench.PixelMajorBF16Workload(lib, *shape, dev, 1)
outs = {}
for rnd in range(2):
    for partial, overlap in ((1, -1), (0, -1), (1, 2), (0, 2), (1, 0)):
        lib.set_option("bf16_partial", partial)
        lib.set_option("planes_overlap", overlap)
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        ms = bench.time_region(wl.step, 20)
        fwd, bwd = bench.time_region(wl.forward, 10), bench.time_region(wl.backward, 10)
        print(f"== round {rnd} bf16 {shape} bf16_partial={partial} planes_overlap={overlap}: step {ms:.4f} ms  fwd {fwd:.4f}  bwd {bwd:.4f}", flush=True)
        outs[partial] = (wl.y.float().clone(), wl.dqkv.float().clone())
        if rnd == 0 and overlap == -1:
            prev = lib.set_option("planes_overlap", 0)
            rec = lib.profile_launches(lambda: [wl.step() for _ in range(3)])
            lib.set_option("planes_overlap", prev)
            n = len(rec) // 3
            for i in range(n):
                print("    %8.1f us  %s" % (sum(rec[r * n + i][1] for r in range(3)) / 3 * 1e3, rec[i][0][:110]))
y1, y0 = outs[1][0], outs[0][0]
g1, g0 = outs[1][1], outs[0][1]
print("bf16 partial vs fp32 partial: max |dy| / |y|max = %.2e, max |d dqkv| / |dqkv|max = %.2e" %
      (float((y1 - y0).abs().max() / y0.abs().max()), float((g1 - g0).abs().max() / g0.abs().max())))
lib.set_option("bf16_partial", 1)
lib.set_option("planes_overlap", -1)
