"""Probe: an fp32 input under bf16 autocast (what RCCAModule hands the attention module in an autocast training run: the output of
InPlaceABNSync is fp32) -- module fwd+bwd on the packed-strips route (stacked conv2d under autocast + the NCHW strip kernels on fp32
copies) against the f32-planes node with autocast switched off inside (its own fp32 / split-bf16 GEMMs + the plane kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ccnet_amd import CrissCrossAttention
from ccnet_amd.functions import CrissCrossPlanesModuleFunction
dev = torch.device("cuda:0")
for B in (1, 2, 8):
    torch.manual_seed(0)
    m = CrissCrossAttention(512).to(dev)
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(B, 512, 97, 97, device=dev, requires_grad=True)
    dy = torch.randn(B, 512, 97, 97, device=dev)

    def cur():
        m.zero_grad(set_to_none=True); x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.backward(dy)
        return y

    def cast():
        m.zero_grad(set_to_none=True); x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            with torch.autocast("cuda", enabled=False):
                params = (m.query_conv.weight, m.query_conv.bias, m.key_conv.weight, m.key_conv.bias, m.value_conv.weight, m.value_conv.bias)
                sg = m.split_bf16_projections and B * 97 * 97 >= m.split_bf16_min_pixels
                y = CrissCrossPlanesModuleFunction.apply(x, *params, m.gamma, sg, False)
        y.backward(dy)
        return y

    with torch.autocast("cuda", dtype=torch.bfloat16):
        route = m.route(x)
    for _ in range(3):
        y0 = cur(); y1 = cast()
    t0, t1 = bench.time_region(cur, 20), bench.time_region(cast, 20)
    print(f"B={B}: fp32 x under autocast, module fwd+bwd: route '{route}' {t0:.3f} ms | f32-planes node, autocast off inside {t1:.3f} ms | "
          f"max |dy| {float((y0 - y1).abs().max()):.1e}", flush=True)
