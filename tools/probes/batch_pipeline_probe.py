#!/usr/bin/env python3
"""Probe: does running the split-plane step as TWO image halves on two streams (each half its own launch chain, so one half's
latency-bound launches -- energies, softmax, dq | dk -- overlap the other half's HBM-bound ones) beat one chain over the
whole batch?  Pure host-side experiment on the shipped entry points; results must be bit-identical.
usage: batch_pipeline_probe.py [B C H W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ccnet_amd import _lib  # noqa: E402

B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 512, 97, 97)
lib = _lib.get_lib()
dev = torch.device("cuda:0")
wl = bench.PlanesWorkload(lib, B, C, H, W, dev, 1234)
cq, ct = C // 8, wl.ct
side = torch.cuda.Stream()
ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()
halves = [(0, B // 2), (B // 2, B - B // 2)]
fws = [torch.empty(lib.ccnet_cca_planes_workspace_bytes(n, C, cq, H, W, 0) // 4 + 64, device=dev) for _, n in halves]
bws = [torch.empty(lib.ccnet_cca_planes_workspace_bytes(n, C, cq, H, W, 1) // 4 + 64, device=dev) for _, n in halves]
dgam = torch.empty(2, device=dev)


def fwd_half(i, stream):
    b0, n = halves[i]
    bs = H * W * ct
    p = wl.qkv[b0:].data_ptr()
    lib.check(lib.ccnet_cca_forward_planes_f32(p, p + 4 * cq, wl.vpl[b0:].data_ptr(), wl.x[b0:].data_ptr(), wl.gamma.data_ptr(),
                                               wl.y[b0:].data_ptr(), wl.A[b0:].data_ptr(), n, C, cq, H, W, bs, ct, bs, ct,
                                               H * W * 2 * C, 2 * C, fws[i].data_ptr(), fws[i].numel() * 4, stream), "fwd half")


def bwd_half(i, stream):
    b0, n = halves[i]
    bs = H * W * ct
    p, g = wl.qkv[b0:].data_ptr(), wl.dqkv[b0:].data_ptr()
    lib.check(lib.ccnet_cca_backward_planes_f32(wl.dy[b0:].data_ptr(), p, p + 4 * cq, wl.vpl[b0:].data_ptr(), wl.A[b0:].data_ptr(),
                                                wl.gamma.data_ptr(), g, g + 4 * cq, g + 8 * cq, dgam[i:].data_ptr(),
                                                wl.scratch[b0:].data_ptr(), n, C, cq, H, W, bs, ct, bs, ct, H * W * 2 * C, 2 * C,
                                                bs, ct, bs, ct, bs, ct, bws[i].data_ptr(), bws[i].numel() * 4, stream), "bwd half")


def two_chains(fn):
    main = torch.cuda.current_stream()
    ev_f.record(main)
    side.wait_event(ev_f)
    fn(0, main.cuda_stream)
    fn(1, side.cuda_stream)
    ev_j.record(side)
    main.wait_event(ev_j)


def step_split():
    two_chains(fwd_half)
    two_chains(bwd_half)


for ov in (-1, 0):
    lib.set_option("planes_overlap", ov)
    for _ in range(5):
        wl.step(); step_split()
    torch.cuda.synchronize()
    wl.step(); torch.cuda.synchronize()
    ref = (wl.y.clone(), wl.dqkv.clone(), wl.A.clone())
    wl.y.zero_(); wl.dqkv.zero_()
    step_split(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(ref, (wl.y, wl.dqkv, wl.A)))
    print(f"planes_overlap={ov}: one chain  step {bench.time_region(wl.step, 30):.4f}  fwd {bench.time_region(wl.forward, 30):.4f}  "
          f"bwd {bench.time_region(wl.backward, 30):.4f} ms")
    print(f"planes_overlap={ov}: two halves step {bench.time_region(step_split, 30):.4f}  "
          f"fwd {bench.time_region(lambda: two_chains(fwd_half), 30):.4f}  bwd {bench.time_region(lambda: two_chains(bwd_half), 30):.4f} ms"
          f"   bit-identical y / dqkv / A: {same}")
lib.set_option("planes_overlap", -1)
