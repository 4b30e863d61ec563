#!/usr/bin/env python3
"""Probe: does running the split-plane FORWARD as several image groups on two streams (each group its own launch chain, so one
group's latency-bound launches -- energies, softmax -- overlap another group's HBM-bound aggregation passes) beat one chain over the
whole batch?  Pure host-side experiment on the shipped entry points (ABI 200, plane-free form); results must be bit-identical.
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


def groups(n):
    k, out, b0 = B // n, [], 0
    for i in range(n):
        nb = k + (1 if i < B - k * n else 0)
        out.append((b0, nb))
        b0 += nb
    return out


def fwd_group(b0, n, ws, stream):
    bs = H * W * ct
    p = wl.qkv[b0:].data_ptr()
    lib.check(lib.ccnet_cca_forward_planes_f32(p, p + 4 * cq, p + 8 * cq, None, None, wl.x[b0:].data_ptr(), wl.gamma.data_ptr(),
                                               wl.y[b0:].data_ptr(), wl.A[b0:].data_ptr(), n, C, cq, H, W, bs, ct, bs, ct, bs, ct,
                                               H * W * 2 * C, 2 * C, ws.data_ptr(), ws.numel() * 4, stream), "fwd group")


def make(n):
    gs = groups(n)
    wss = [torch.empty(lib.ccnet_cca_planes_workspace_bytes(nb, C, cq, H, W, 0) // 4 + 64, device=dev) for _, nb in gs]

    def run():
        main = torch.cuda.current_stream()
        ev_f.record(main)
        side.wait_event(ev_f)
        for i, (b0, nb) in enumerate(gs):                    # groups alternate between the two streams
            fwd_group(b0, nb, wss[i], (main if i % 2 == 0 else side).cuda_stream)
        ev_j.record(side)
        main.wait_event(ev_j)
    return run


for _ in range(5):
    wl.forward()
torch.cuda.synchronize()
ref = (wl.y.clone(), wl.A.clone())
g0 = bench.capture_step_graph(wl.forward)          # (replayed graphs: the eager numbers carry the host's launch overhead, 8 .. 32 launches)
print(f"one chain over the batch: fwd eager {bench.time_region(wl.forward, 30):.4f} ms, graph replay {bench.time_region(g0.replay, 50):.4f} ms")
for n in (2, 4, 8):
    run = make(n)
    wl.y.zero_()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    same = torch.equal(ref[0], wl.y) and torch.equal(ref[1], wl.A)
    g = bench.capture_step_graph(run)
    print(f"{n} image groups on two streams: fwd eager {bench.time_region(run, 30):.4f} ms, graph replay {bench.time_region(g.replay, 50):.4f} ms"
          f"   bit-identical y / A: {same}")
    del g
