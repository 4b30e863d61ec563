#!/usr/bin/env python3
"""Run the step (or one stage / branch) a few times -- the command rocprofv3 wraps for PMC passes.
usage: python tools/run_stage.py [--lib path.so] [--iters N] [--family planes|strips] [--stage NAME --mask 1|2|3]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ccnet_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=_lib.LIB_PATH)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--stage", default="")
ap.add_argument("--mask", type=int, default=3)
ap.add_argument("--family", default="planes", choices=("planes", "strips"), help="planes = the bench's default f32 workload")
a = ap.parse_args()
lib = _lib.CcaLibrary(os.path.join(ROOT, a.lib))
cls = bench.PlanesWorkload if a.family == "planes" and not a.stage else bench.CoreWorkload
wl = cls(lib, 8, 512, 97, 97, torch.device("cuda:0"), 1234)
wl.step()
torch.cuda.synchronize()
if a.stage:
    fn = wl.stage_table()[a.stage][0]
    lib.ccnet_cca_set_branch_mask(a.mask)
    for _ in range(a.iters):
        lib.check(fn())
else:
    for _ in range(a.iters):
        wl.step()
torch.cuda.synchronize()
