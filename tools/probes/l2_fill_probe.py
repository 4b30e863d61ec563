#!/usr/bin/env python3
"""Round 5, box classes: the 25 KB LDS-DMA tile fill out of a source that LIVES IN THE L2 (0.5 MB and 2 MB: every XCD's 4 MiB L2 holds
it after the first touch), rows 256 B and 2560 B apart, alone and with 768 workgroups -- the L2-hit side of the fills whose HBM side is
in bench.py's gpu_probe.  (profiles/r05a_slow_box_bench.json has this quantity for a slow box by accident: its probe build re-used
0.55 MB of rows -- 25.5 TB/s, 0.63 us per fill loaded, 0.40 us alone.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccnet_amd import _lib  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
print(f"{'source':>8} {'row stride':>11} {'alone us':>9} {'cycles':>7} {'768 wg us':>10} {'cycles':>7} {'TB/s':>7}")
for mb in (0.5, 2.0):
    src = torch.empty(int(mb * 1024 * 1024 / 4), device=dev).normal_()
    for stride in (256, 2560):
        row = []
        for n, reps in ((1, 2000), (768, 1000)):
            ck = torch.zeros(n * 4, dtype=torch.int64, device=dev)
            for _ in range(3):
                lib.check(lib.ccnet_cca_probe_dma(src.data_ptr(), src.numel() * 4, ck.data_ptr(), n, reps, stride, st), "probe_dma")
            torch.cuda.synchronize()
            k = ck.cpu().numpy().reshape(n, 4).astype("float64")
            span = (k[:, 3].max() - k[:, 2].min()) * 1e-8
            row += [float(((k[:, 3] - k[:, 2]) / reps).mean()) * 1e-2, float((k[:, 0] / reps).mean()), n * reps * 25600.0 / span / 1e12]
        print(f"{mb:6.1f}MB {stride:11d} {row[0]:9.2f} {row[1]:7.0f} {row[3]:10.2f} {row[4]:7.0f} {row[5]:7.2f}")
