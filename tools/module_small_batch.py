"""module fwd+bwd (projections + core + autograd) at several batch sizes, NCHW fp32 tensors: the reference-shaped strip route vs the
split-plane node with fp32 and with split-bf16 projection GEMMs (the default at every size since round 6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ccnet_amd import CrissCrossAttention
dev = torch.device("cuda:0")
C, H, W = 512, 97, 97
for B in (1, 2, 4, 8):
    row, ys = [], []
    # (reference-shaped fallback: three convolutions + NCHW strip kernels | split-plane node, fp32 GEMMs | split-plane node, split-bf16 GEMMs)
    for fuse, sg in ((False, False), (True, False), (True, True)):
        torch.manual_seed(0)
        m = CrissCrossAttention(C).to(dev)
        m.fuse_projections, m.split_bf16_projections = fuse, sg
        m.split_bf16_min_pixels = 0 if sg else 10 ** 9
        with torch.no_grad():
            m.gamma.fill_(0.5)
        x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
        dy = torch.randn(B, C, H, W, device=dev)
        def one():
            m.zero_grad(set_to_none=True); x.grad = None
            y = m(x); y.backward(dy); return y
        for _ in range(3): y = one()
        torch.cuda.synchronize()
        ys.append((y.detach().clone(), x.grad.clone(), m.value_conv.weight.grad.clone()))
        row.append(bench.time_region(one, 20))
    d = [float((a - b).abs().max()) for a, b in zip(ys[0], ys[1])]
    d3 = [float((a - b).abs().max()) for a, b in zip(ys[0], ys[2])]
    print(f"B={B}: module fwd+bwd  separate-strips route (three convolutions + NCHW strip kernels) {row[0]:.3f} ms | split-plane node, fp32 GEMMs "
          f"{row[1]:.3f} ms | split-plane node + split-bf16 GEMMs {row[2]:.3f} ms   (max |diff| vs the strip route: fp32 GEMMs y {d[0]:.1e} dx "
          f"{d[1]:.1e} dWv {d[2]:.1e}; split GEMMs y {d3[0]:.1e} dx {d3[1]:.1e} dWv {d3[2]:.1e})", flush=True)

# ---- round 4: the default module eager vs captured into two hipGraphs (ccnet_amd.graph_module), and the GPU time of its launches ----
from ccnet_amd import graph_module
from ccnet_amd import _lib as _L
lib = _L.get_lib()
for B in (1, 2, 4):
    torch.manual_seed(0)
    m = CrissCrossAttention(C).to(dev)
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    dy = torch.randn(B, C, H, W, device=dev)

    def one(f):
        y = f(x)
        y.backward(dy)

    for _ in range(3):
        one(m)
    t_eager = bench.time_region(lambda: one(m), 30)
    rec = lib.profile_launches(lambda: one(m))      # (before graph_module: make_graphed_callables replaces m.forward)
    torch.cuda.synchronize()
    g = graph_module(m, x.detach().clone().requires_grad_(True))
    for _ in range(3):
        one(g)
    t_graph = bench.time_region(lambda: one(g), 30)
    print(f"B={B}: default module fwd+bwd eager {t_eager:.3f} ms | graphed (ccnet_amd.graph_module) {t_graph:.3f} ms | the library's own "
          f"{len(rec)} launches sum to {sum(t for _, t in rec):.3f} ms (the projection GEMMs among them since round 6; the reductions around them are torch's)", flush=True)
