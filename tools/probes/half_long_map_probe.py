"""Probe: bf16 module inference on maps beyond the bf16 kernels' 132 positions: route f32-planes-cast (the blocked fp32 plane kernels on
fp32 copies) against the packed-strips route it replaced there (windowed / any-shape strip kernels through fp32 copies)."""
import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ccnet_amd import CrissCrossAttention
dev = torch.device("cuda:0")
for shape in ((1, 512, 129, 257), (1, 512, 257, 513)):
    torch.manual_seed(0)
    m = CrissCrossAttention(shape[1]).to(dev).to(torch.bfloat16).eval()
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(*shape, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        r1 = m.route(x); y1 = m(x); t1 = bench.time_region(lambda: m(x), 10)
        m.split_planes = False
        r0 = m.route(x); y0 = m(x); t0 = bench.time_region(lambda: m(x), 3)
    print(f"{shape} bf16 inference: route '{r1}' {t1:.3f} ms | route '{r0}' {t0:.3f} ms | max |diff| {float((y1.float() - y0.float()).abs().max()):.1e}", flush=True)
