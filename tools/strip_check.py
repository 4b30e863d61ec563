"""GPU check of the strip kernels' entry points against fp64 einsum references (run-to-run identity included)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from ccnet_amd import _lib
lib = _lib.CcaLibrary(sys.argv[1]) if len(sys.argv) > 1 else _lib.get_lib()
dev = torch.device('cuda'); s = torch.cuda.current_stream().cuda_stream
def run(B, C, H, W):
    g = torch.Generator().manual_seed(1)
    q, k = (torch.randn(B, C // 8, H, W, generator=g).to(dev) for _ in range(2))
    v, x, dy = (torch.randn(B, C, H, W, generator=g).to(dev) for _ in range(3))
    gamma = torch.full((1,), 0.5, device=dev)
    A = torch.empty(B, H, W, H + W, device=dev)
    lib.check(lib.ccnet_ca_forward_f32(q.data_ptr(), k.data_ptr(), A.data_ptr(), B, C // 8, H, W, 1, s))
    e = torch.einsum('bchw,bcjw->bhwj', q.double(), k.double()); e[:, torch.arange(H), :, torch.arange(H)] = float('-inf')
    Ar = torch.softmax(torch.cat([e, torch.einsum('bchw,bchj->bhwj', q.double(), k.double())], 3), 3)
    res = {'A': float((A - Ar).abs().max())}
    A = Ar.float()
    y = torch.full_like(x, float('nan')); y2 = torch.full_like(x, float('nan'))
    lib.check(lib.ccnet_ca_map_forward_f32(A.data_ptr(), v.data_ptr(), x.data_ptr(), gamma.data_ptr(), y.data_ptr(), B, C, H, W, s))
    lib.check(lib.ccnet_ca_map_forward_f32(A.data_ptr(), v.data_ptr(), x.data_ptr(), gamma.data_ptr(), y2.data_ptr(), B, C, H, W, s))
    ref = 0.5 * (torch.einsum('bhwj,bcjw->bchw', Ar[..., :H], v.double()) + torch.einsum('bhwj,bchj->bchw', Ar[..., H:], v.double())) + x.double()
    res['y'] = float((y - ref).abs().max()); res['y_rr'] = float((y - y2).abs().max())
    dA = torch.full_like(A, float('nan')); dv = torch.full_like(v, float('nan'))
    lib.check(lib.ccnet_ca_map_backward_f32(dy.data_ptr(), A.data_ptr(), v.data_ptr(), gamma.data_ptr(), dA.data_ptr(), dv.data_ptr(), B, C, H, W, s))
    dvr = 0.5 * (torch.einsum('bhwj,bchw->bcjw', Ar[..., :H], dy.double()) + torch.einsum('bhwj,bchw->bchj', Ar[..., H:], dy.double()))
    tr = torch.cat([torch.einsum('bchw,bcjw->bhwj', dy.double(), v.double()), torch.einsum('bchw,bchj->bhwj', dy.double(), v.double())], 3)
    res['dv'] = float((dv - dvr).abs().max()); res['dA'] = float((dA - tr).abs().max())
    dq = torch.full_like(q, float('nan')); dk = torch.full_like(k, float('nan'))
    dE = torch.randn(B, H, W, H + W, generator=g).to(dev)
    lib.check(lib.ccnet_ca_backward_f32(dE.data_ptr(), q.data_ptr(), k.data_ptr(), dq.data_ptr(), dk.data_ptr(), B, C // 8, H, W, s))
    dqr = torch.einsum('bhwj,bcjw->bchw', dE[..., :H].double(), k.double()) + torch.einsum('bhwj,bchj->bchw', dE[..., H:].double(), k.double())
    dkr = torch.einsum('bhwj,bchw->bcjw', dE[..., :H].double(), q.double()) + torch.einsum('bhwj,bchw->bchj', dE[..., H:].double(), q.double())
    res['dq'] = float((dq - dqr).abs().max()); res['dk'] = float((dk - dkr).abs().max())
    torch.cuda.synchronize()
    print((B, C, H, W), {k_: f'{v_:.2e}' for k_, v_ in res.items()}, flush=True)
for shp in ((1, 64, 97, 97), (2, 128, 96, 96), (2, 128, 65, 80), (2, 128, 49, 100), (1, 64, 33, 40)):
    run(*shp)
