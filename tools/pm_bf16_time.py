"""Time the pixel-major core (csrc/cca_gmap.hpp) fwd+bwd through the C ABI: python tools/pm_bf16_time.py [B C H W [bf16|f32]]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import _lib

B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 512, 129, 129)
F32 = len(sys.argv) > 5 and sys.argv[5] == "f32"
dt, es = (torch.float32, 4) if F32 else (torch.bfloat16, 2)
cq, dev = C // 8, torch.device("cuda:0")
L = _lib.get_lib()
ct = C + 2 * cq
g = torch.Generator(device="cpu").manual_seed(1)
qkv = (torch.randn(B, H, W, ct, generator=g) * 0.5).to(dev).to(dt)
x = torch.randn(B, H, W, C, generator=g).to(dev).to(dt)
dy = torch.randn(B, H, W, C, generator=g).to(dev).to(dt)
gamma = torch.tensor([0.5], device=dev)
y, dqkv = torch.empty_like(x), torch.empty_like(qkv)
A = torch.empty(B, H, W, H + W, device=dev)
scr = torch.empty_like(A)
dg = torch.empty(1, device=dev)
nf, nb = L.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 0), L.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 1)
ws = torch.empty(max(nf, nb) // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
p, gq, bs = qkv.data_ptr(), dqkv.data_ptr(), H * W * ct


def fwd():
    L.check((L.ccnet_cca_forward_pm_f32 if F32 else L.ccnet_cca_forward_pm_bf16)(p, p + es * cq, p + 2 * es * cq, x.data_ptr(), gamma.data_ptr(), y.data_ptr(), A.data_ptr(),
                                        B, C, cq, H, W, bs, ct, bs, ct, bs, ct, H * W * C, C, H * W * C, C, ws.data_ptr(), nf, st))


def bwd():
    L.check((L.ccnet_cca_backward_pm_f32 if F32 else L.ccnet_cca_backward_pm_bf16)(dy.data_ptr(), p, p + es * cq, p + 2 * es * cq, A.data_ptr(), gamma.data_ptr(), gq, gq + es * cq,
                                         gq + 2 * es * cq, dg.data_ptr(), scr.data_ptr(), B, C, cq, H, W, H * W * C, C, bs, ct, bs, ct,
                                         bs, ct, bs, ct, bs, ct, bs, ct, ws.data_ptr(), nb, st))


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tf, tb = timeit(fwd), timeit(bwd)
# algorithmic bytes: bf16 q,k,v,x,y + dy,dq,dk,dv; fp32 A written once + read (fwd), read twice + dA/dE traffic
feat = B * H * W * es
alg = feat * (2 * cq + C + C + C) + feat * (C + 2 * cq + C + 2 * cq + C) + 2 * B * H * W * (H + W) * 4
print(f"pm {'f32' if F32 else 'bf16'} ({B},{C},{H},{W}): fwd {tf:.3f} ms  bwd {tb:.3f} ms  total {tf + tb:.3f} ms ; minimal bytes {alg / 1e9:.3f} GB "
      f"-> {alg / (tf + tb) / 1e6:.0f} GB/s")
if len(sys.argv) < 7:
    sys.exit(0)
# compare with the fp32 strip path on the same values
from ccnet_amd import criss_cross_attention
nchw = lambda t: t.permute(0, 3, 1, 2).float().contiguous()
q, k, v = nchw(qkv[..., :cq]), nchw(qkv[..., cq:2 * cq]), nchw(qkv[..., 2 * cq:])
xs, dys = nchw(x), nchw(dy)
leaves = [t.requires_grad_(True) for t in (q, k, v, xs)]
def f32():
    yy = criss_cross_attention(*leaves, gamma)
    yy.backward(dys)
print("fp32 strip path fwd+bwd (autograd): %.3f ms" % timeit(f32, 5))
