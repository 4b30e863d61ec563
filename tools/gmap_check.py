"""GPU check + timing of the pixel-major strip kernels (csrc/cca_gmap.hpp) against einsum references and the NCHW strip launches."""
import sys, torch
sys.path.insert(0, '/root/repo')
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device('cuda'); s = torch.cuda.current_stream().cuda_stream
def tm(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
def run(B, C, H, W):
    g = torch.Generator().manual_seed(1)
    A = torch.softmax(torch.randn(B, H, W, H + W, generator=g) * 3, -1).to(dev)
    v, dy = (torch.randn(B, C, H, W, generator=g).to(dev) for _ in range(2))
    gamma = torch.full((1,), 0.5, device=dev)
    vpm, dypm = v.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
    o1, o2 = torch.full_like(vpm, float('nan')), torch.full_like(vpm, float('nan'))
    P = lambda t: t.data_ptr()
    call = lambda T, F, add, gm, out, row, tr: lib.check(lib.ccnet_ca_strip_map_pm_f32(P(T), P(F), P(add) if add is not None else None, P(gm) if gm is not None else None, P(out), B, C, H, W, H * W * C, C, H * W * C, C, row, tr, s))
    call(A, vpm, None, None, o1, 0, 0); call(A, vpm, o1, None, o2, 1, 0)
    ref = torch.einsum('bhwj,bcjw->bchw', A[..., :H].double(), v.double()) + torch.einsum('bhwj,bchj->bchw', A[..., H:].double(), v.double())
    e_f = float((o2.permute(0, 3, 1, 2) - ref).abs().max())
    call(A, dypm, None, gamma, o1, 0, 1); call(A, dypm, o1, gamma, o2, 1, 1)
    dvr = 0.5 * (torch.einsum('bhwj,bchw->bcjw', A[..., :H].double(), dy.double()) + torch.einsum('bhwj,bchw->bchj', A[..., H:].double(), dy.double()))
    e_t = float((o2.permute(0, 3, 1, 2) - dvr).abs().max())
    t = {n: tm(lambda a=a: call(*a)) for n, a in (("col", (A, vpm, None, None, o1, 0, 0)), ("row+add", (A, vpm, o1, None, o2, 1, 0)),
                                                    ("colT", (A, dypm, None, gamma, o1, 0, 1)), ("rowT+add", (A, dypm, o1, gamma, o2, 1, 1)))}
    x = torch.randn_like(v); y = torch.empty_like(v); dv = torch.empty_like(v)
    ts = {}
    for mask, nm in ((1, "col"), (2, "row")):
        lib.ccnet_cca_set_branch_mask(mask)
        ts["nchw " + nm + " fwd(+x)"] = tm(lambda: lib.check(lib.ccnet_ca_map_forward_f32(P(A), P(v), P(x), P(gamma), P(y), B, C, H, W, s)))
        ts["nchw " + nm + " dv"] = tm(lambda: lib.check(lib.ccnet_ca_map_backward_f32(P(dy), P(A), P(v), P(gamma), None, P(dv), B, C, H, W, s)))
    lib.ccnet_cca_set_branch_mask(3)
    print((B, C, H, W), "err fwd %.2e  dv %.2e" % (e_f, e_t), "| pm us:", t, "|", ts, flush=True)
for shp in ((1, 64, 33, 18), (8, 512, 97, 97), (8, 512, 65, 65), (1, 512, 97, 97)):
    run(*shp)
