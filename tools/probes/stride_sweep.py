#!/usr/bin/env python3
"""Round 5, box classes: how does the 25 KB LDS-DMA tile fill (ccnet_cca_probe_dma: 100 rows of 256 B) depend on the
distance between its rows?  256 B = one contiguous piece; 2560 B = a ROW strip of the packed fp32 projection (a pixel's 64 channels every
640 floats); 248 320 B = a COLUMN strip of it at W = 97; 4 KiB .. 2 MiB = one row per page of that size.  Per stride: time per fill
alone and with 768 workgroups streaming, and the rate they stream at, out of a 1.5 GiB source (every fill misses the L2).  If the
fill time rises with the stride, address translation (one page-table walk per row) is what a column strip pays for."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccnet_amd import _lib  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
src = torch.empty(400 * 1024 * 1024, device=dev).normal_()          # 1.6 GB (the entry point clamps views to 2 GiB)
st = torch.cuda.current_stream().cuda_stream
print(f"{'row stride':>12} {'alone us':>9} {'cycles':>8} {'768 wg us':>10} {'cycles':>8} {'TB/s':>7}")
for stride in (256, 2560, 4096, 16384, 65536, 97 * 640 * 4, 1 << 20, 2 << 20, 8 << 20):
    row = []
    for n, reps in ((1, 300), (768, 200)):
        ck = torch.zeros(n * 4, dtype=torch.int64, device=dev)
        rc = 0
        for _ in range(2):
            rc = lib.ccnet_cca_probe_dma(src.data_ptr(), src.numel() * 4, ck.data_ptr(), n, reps, stride, st)
        torch.cuda.synchronize()
        if rc != 0:
            row += [float("nan")] * 3
            continue
        k = ck.cpu().numpy().reshape(n, 4).astype("float64")
        span = (k[:, 3].max() - k[:, 2].min()) * 1e-8
        row += [float(((k[:, 3] - k[:, 2]) / reps).mean()) * 1e-2, float((k[:, 0] / reps).mean()), n * reps * 25600.0 / span / 1e12]
    print(f"{stride:12d} {row[0]:9.2f} {row[1]:8.0f} {row[3]:10.2f} {row[4]:8.0f} {row[5]:7.2f}")
