"""GPU check of the row-band forward kernel against the strip kernels (same A), plus timing."""
import sys, torch
sys.path.insert(0, '/root/repo')
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device('cuda'); s = torch.cuda.current_stream().cuda_stream
def run(B, C, H, W, iters=20):
    g = torch.Generator().manual_seed(1)
    q, k = (torch.randn(B, C // 8, H, W, generator=g).to(dev) for _ in range(2))
    v, x = (torch.randn(B, C, H, W, generator=g).to(dev) for _ in range(2))
    gamma = torch.full((1,), 0.5, device=dev)
    A = torch.empty(B, H, W, H + W, device=dev)
    lib.check(lib.ccnet_ca_forward_f32(q.data_ptr(), k.data_ptr(), A.data_ptr(), B, C // 8, H, W, 1, s))
    y0, y1 = torch.empty_like(x), torch.full_like(x, float('nan'))
    vpm = v.permute(0, 2, 3, 1).contiguous()
    f0 = lambda: lib.check(lib.ccnet_ca_map_forward_f32(A.data_ptr(), v.data_ptr(), x.data_ptr(), gamma.data_ptr(), y0.data_ptr(), B, C, H, W, s))
    f1 = lambda: lib.check(lib.ccnet_ca_map_forward_pm_f32(A.data_ptr(), vpm.data_ptr(), x.data_ptr(), gamma.data_ptr(), y1.data_ptr(), B, C, H, W, H * W * C, C, s))
    f0(); f1(); torch.cuda.synchronize()
    ref = (gamma.double() * (torch.einsum('bhwj,bcjw->bchw', A[..., :H].double(), v.double()) + torch.einsum('bhwj,bchj->bchw', A[..., H:].double(), v.double())) + x.double())
    print((B, C, H, W), 'strip err', float((y0 - ref).abs().max()), 'band err', float((y1 - ref).abs().max()), flush=True)
    for name, f in (('strip', f0), ('band', f1)):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); e1.synchronize()
        print('   ', name, round(e0.elapsed_time(e1) / iters * 1e3, 1), 'us', flush=True)
for shp in ((1, 32, 33, 18), (1, 32, 97, 20), (2, 512, 97, 97), (8, 512, 97, 97), (8, 512, 65, 65), (8, 512, 96, 96), (1, 512, 97, 97)):
    run(*shp)
