#!/usr/bin/env python3
"""A/B of the fp32 core families at one shape on this GPU: NCHW strips vs split planes under its options (ring kernels, persistent dA, dv on the side stream).
Per family: fwd+bwd ms (eager, HIP events), the per-launch durations inside a step (library launch profiler) and the
agreement of the plane variants with each other on the same inputs.
usage: family_compare.py [B C H W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ccnet_amd import _lib  # noqa: E402

B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 512, 97, 97)
lib = _lib.get_lib()
dev = torch.device("cuda:0")
fams = [("nchw-strips", bench.CoreWorkload), ("planes-noring", bench.PlanesWorkload), ("planes-3wg-nostream", bench.PlanesWorkload),
        ("planes-3wg-nooverlap", bench.PlanesWorkload), ("planes-3wg-overlap1", bench.PlanesWorkload),
        ("planes-3wg", bench.PlanesWorkload)]
RING = {"planes-noring": 0, "planes-3wg": 2, "planes-3wg-nostream": 2, "planes-3wg-nooverlap": 2, "planes-3wg-overlap1": 2}


def set_options(name):
    lib.set_option("planes_ring", RING.get(name, 2))
    lib.set_option("planes_stream", 0 if name.endswith("nostream") else 1)
    lib.set_option("planes_overlap", 0 if name.endswith("nooverlap") or name.endswith("noring") else 1 if name.endswith("overlap1") else -1)


res = {}
for name, cls in fams:
    set_options(name)
    wl = cls(lib, B, C, H, W, dev, 1234)
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    ms = bench.time_region(wl.step, 30)
    fwd, bwd = bench.time_region(wl.forward, 20), bench.time_region(wl.backward, 20)
    rec = lib.profile_launches(lambda: [wl.step() for _ in range(5)])
    n = len(rec) // 5
    print(f"== {name} ({B},{C},{H},{W}): step {ms:.4f} ms  fwd {fwd:.4f}  bwd {bwd:.4f}  launches {n}  "
          f"event sum {sum(t for _, t in rec) / 5:.4f}")
    for i in range(n):
        print(f"     {sum(rec[r * n + i][1] for r in range(5)) / 5 * 1e3:8.1f} us  {rec[i][0][:110]}")
    res[name] = wl
b = res["planes-noring"]
set_options("planes-noring")
b.step()
for fam, ring in RING.items():
    a = res[fam]
    set_options(fam)
    a.step()
    torch.cuda.synchronize()
    for nm in ("y", "dqkv", "A", "dgamma"):
        d = (getattr(a, nm) - getattr(b, nm)).abs().max().item()
        print(f"{fam} vs planes-noring: max |d {nm}| = {d:.3e}   (max |ref| {getattr(b, nm).abs().max().item():.3e})")
set_options("planes-3wg")      # the shipped defaults
