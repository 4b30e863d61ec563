#!/usr/bin/env python3
"""Which class of box is this?  The NCHW strip family's headline step reads 0.98-1.00 ms on a normal box of the pool and 1.16-1.28 ms on
its slow ones (profiles/r04b_*, r05a_*): a ten-second classifier.  Prints 'slow' or 'normal' + the numbers; exit code 0 / 1 = slow / normal."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ccnet_amd import _lib  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
wl = bench.CoreWorkload(lib, 8, 512, 97, 97, dev, 1234)
for _ in range(30):
    wl.step()
torch.cuda.synchronize()
strips = bench.time_region(wl.step, 40)
pl = bench.PlanesWorkload(lib, 8, 512, 97, 97, dev, 1234)
for _ in range(30):
    pl.step()
torch.cuda.synchronize()
planes = bench.time_region(pl.step, 40)
cls = "slow" if strips >= 1.1 else "normal"
print(f"{cls}: strip family {strips:.4f} ms, split-plane step {planes:.4f} ms (eager)")
sys.exit(0 if cls == "slow" else 1)
