"""Repeat the pixel-major bf16 core and the split-plane fp32 core (NCHW x / y / dy) under concurrent HBM load and count runs whose
outputs differ bit-wise from the first run -- a race detector for the counted-vmcnt pipelines and the register-prefetched
epilogues.  usage: stress_pm.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
noise = torch.randn(64 * 1024 * 1024, device=dev)
side = torch.cuda.Stream()
cases = [("bf16 (16,512,129,129)", bench.PixelMajorBF16Workload(lib, 16, 512, 129, 129, dev, 7)),
         ("bf16 (2,512,97,97)", bench.PixelMajorBF16Workload(lib, 2, 512, 97, 97, dev, 8)),
         ("f32 planes (8,512,97,97)", bench.PlanesWorkload(lib, 8, 512, 97, 97, dev, 9)),
         ("f32 planes (1,512,97,97)", bench.PlanesWorkload(lib, 1, 512, 97, 97, dev, 10)),
         ("f32 planes (3,256,100,61)", bench.PlanesWorkload(lib, 3, 256, 100, 61, dev, 11)),
         ("f32 planes, long rows (2,512,129,257)", bench.PlanesWorkload(lib, 2, 512, 129, 257, dev, 12)),       # blocks of <= 100
         ("f32 planes, long rows (1,128,60,500)", bench.PlanesWorkload(lib, 1, 128, 60, 500, dev, 13)),          # blocks of <= 132
         ("f32 planes, long columns and rows (1,256,161,321)", bench.PlanesWorkload(lib, 1, 256, 161, 321, dev, 14)),   # blocked column passes too
         ("f32 planes, long columns (1,128,402,97)", bench.PlanesWorkload(lib, 1, 128, 402, 97, dev, 15))]       # 132-position column blocks
for name, wl in cases:
    outs = lambda: (wl.y, wl.dqkv, wl.dgamma, wl.A)          # noqa: E731
    wl.step(); torch.cuda.synchronize()
    ref = [t.clone() for t in outs()]
    assert all(torch.isfinite(t.float()).all() for t in ref), name
    bad = [0, 0, 0, 0]
    for i in range(iters):
        if i % 2:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        wl.step()
        torch.cuda.synchronize()
        for j, (a, b) in enumerate(zip(outs(), ref)):
            bad[j] += int(not torch.equal(a, b))
    print(f"{name}: {iters} iterations, runs that differ from run 0: y {bad[0]} dqkv {bad[1]} dgamma {bad[2]} A {bad[3]}", flush=True)


# ---- round 6: the three-plane backward (ccnet_cca_backward_planes3_f32: three 8-byte stores per pixel row + the column-sum row of
# ---- wavefront 0 under the dv row pass's counted barriers) against the fp32 backward's outputs, repeated under the same load
wl = bench.PlanesWorkload(lib, 8, 512, 97, 97, dev, 9)
B, C, H, W = wl.shape
cq, ct, hw = C // 8, C + 2 * (C // 8), H * W
wl.step(); torch.cuda.synchronize()
hi = wl.dqkv.to(torch.bfloat16)
lo = (wl.dqkv - hi.float()).to(torch.bfloat16)
ref = torch.stack([hi, lo, hi], dim=3).view(B, H, W, 3, ct)
d3 = torch.empty((B, H, W, 3, ct), device=dev, dtype=torch.bfloat16)
db = torch.empty((ct,), device=dev)
n = lib.ccnet_cca_workspace_bytes(_lib.CCNET_WS_PLANES3_BACKWARD, B, C, cq, H, W)
ws = torch.empty(n // 4 + 64, device=dev)
p, bs = wl.qkv.data_ptr(), hw * ct
bad, db0 = [0, 0], None
for i in range(iters):
    if i % 2:
        with torch.cuda.stream(side):
            noise.mul_(1.0001)
    d3.fill_(float("nan"))
    lib.check(lib.ccnet_cca_backward_planes3_f32(wl.dy.data_ptr(), p, p + 4 * cq, p + 8 * cq, wl.A.data_ptr(), wl.gamma.data_ptr(), d3.data_ptr(),
                                                 db.data_ptr(), wl.dgamma.data_ptr(), wl.scratch.data_ptr(), B, C, cq, H, W, bs, ct, bs, ct, bs, ct,
                                                 hw * 3 * ct, 3 * ct, ws.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "planes3")
    torch.cuda.synchronize()
    db0 = db.clone() if db0 is None else db0
    bad[0] += int(not torch.equal(d3, ref))
    bad[1] += int(not torch.equal(db, db0))
print(f"f32 planes, three-plane backward (8,512,97,97): {iters} iterations, runs whose planes differ from the split of the fp32 gradients: {bad[0]}, "
      f"whose bias gradients differ from run 0: {bad[1]}", flush=True)


# ---- round 6: the projection GEMMs (csrc/cca_gemm.hpp: counted LDS-DMA barriers, hand-counted LDS read waits, slab partials added
# ---- in a fixed order) at the module's shapes and at ragged ones, repeated under the same load: outputs bit-identical to run 0
from ccnet_amd import functions as F  # noqa: E402
g = torch.Generator(device=dev).manual_seed(21)
r16 = lambda *shape: torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)  # noqa: E731
for (B, C, H, W) in ((8, 512, 97, 97), (3, 288, 45, 67)):      # (the second: tails in M, N, K and in the slabs)
    hw, ct = H * W, C + 2 * (C // 8)
    x3, w3, w3t, d3 = r16(B * hw, 3 * C), r16(ct, 3 * C), r16(C, 3 * ct), r16(B, hw, 3 * ct)
    dy, bias = torch.randn((B, C, hw), device=dev, generator=g), torch.randn((ct,), device=dev, generator=g)
    run = lambda: (F._projection_gemm(lib, x3, w3, bias), F._projection_adjoint_gemm(lib, w3t, d3, dy),  # noqa: E731
                   F._projection_wgrad_gemm(lib, d3.view(B * hw * 3, ct), x3.view(B * hw * 3, C)))
    ref = run(); torch.cuda.synchronize()
    assert all(t is not None and bool(torch.isfinite(t).all()) for t in ref)
    bad = [0, 0, 0]
    for i in range(iters):
        if i % 2:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        out = run()
        torch.cuda.synchronize()
        for j, (a, b) in enumerate(zip(out, ref)):
            bad[j] += int(not torch.equal(a, b))
    print(f"projection GEMMs ({B},{C},{H},{W}): {iters} iterations, runs that differ from run 0: forward {bad[0]} dx {bad[1]} dW {bad[2]}", flush=True)
