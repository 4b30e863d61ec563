"""Where does a (B,512,97,97) module fwd+bwd spend its time at 1-2 images per GPU?

Prints, per batch size: the eager step (events), the host-side issue time of the same step (wall clock until the last launch
is queued, no synchronisation), the step replayed from ONE manually captured hipGraph holding forward + backward on static
buffers (no copies: hipGraph's best case), and `ccnet_amd.graph_module` (torch.cuda.make_graphed_callables).
`--eager-only N` runs N eager steps and nothing else: the mode `rocprofv3 --kernel-trace --stats` is pointed at to read the GPU
kernel sum of a step (TotalDurationNs of all kernels / N)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from ccnet_amd import CrissCrossAttention, graph_module

ap = argparse.ArgumentParser()
ap.add_argument("--eager-only", type=int, default=0)
ap.add_argument("--batches", default="1,2")
args = ap.parse_args()
dev = torch.device("cuda:0")
C, H, W = 512, 97, 97


def make(B):
    torch.manual_seed(0)
    m = CrissCrossAttention(C).to(dev)
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    dy = torch.randn(B, C, H, W, device=dev)
    return m, x, dy


for B in [int(b) for b in args.batches.split(",")]:
    m, x, dy = make(B)

    def step(f=m):
        y = f(x)
        y.backward(dy)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    if args.eager_only:
        for _ in range(args.eager_only):
            step()
        torch.cuda.synchronize()
        print(f"B={B}: {args.eager_only} eager steps done (kernel sum = rocprofv3's total / {args.eager_only + 5})", flush=True)
        continue
    t_eager = bench.time_region(step, 50)
    # host issue time: queue 50 steps back to back, stop the clock before waiting for the GPU
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    t_issue = (time.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()

    # ONE hipGraph with forward + backward on static tensors
    for p in m.parameters():
        p.grad = None
    x.grad = None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    for p in m.parameters():
        p.grad = None
    x.grad = None
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    t_one = bench.time_region(g.replay, 50)
    del g

    m2, x2, dy2 = make(B)
    gm = graph_module(m2, x2.detach().clone().requires_grad_(True))

    def gstep():
        y = gm(x2)
        y.backward(dy2)

    for _ in range(3):
        gstep()
    t_gm = bench.time_region(gstep, 50)
    print(f"B={B}: eager {t_eager:.3f} ms (host issue {t_issue:.3f} ms per step) | ONE manual hipGraph fwd+bwd {t_one:.3f} ms | "
          f"graph_module {t_gm:.3f} ms", flush=True)
