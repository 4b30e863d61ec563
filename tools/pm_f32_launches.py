#!/usr/bin/env python3
"""Per-launch durations of the all-pixel-major fp32 core (x / y / dy pixel-major as well: what the module's channels_last route runs)
at a shape -- the NCHW tax of the split-plane step's row pass is the difference to its pixel-major sibling.
usage: python tools/pm_f32_launches.py [B C H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccnet_amd import _lib
B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 512, 97, 97)
cq, dev = C // 8, torch.device("cuda:0")
L = _lib.get_lib()
ct = C + 2 * cq
g = torch.Generator(device="cpu").manual_seed(1)
qkv = (torch.randn(B, H, W, ct, generator=g) * 0.5).to(dev)
x = torch.randn(B, H, W, C, generator=g).to(dev)
dy = torch.randn(B, H, W, C, generator=g).to(dev)
gamma = torch.tensor([0.5], device=dev)
y, dqkv = torch.empty_like(x), torch.empty_like(qkv)
A = torch.empty(B, H, W, H + W, device=dev)
scr = torch.empty_like(A)
dg = torch.empty(1, device=dev)
nf, nb = L.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 0), L.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 1)
ws = torch.empty(max(nf, nb) // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
p, gq, bs = qkv.data_ptr(), dqkv.data_ptr(), H * W * ct


def step():
    L.check(L.ccnet_cca_forward_pm_f32(p, p + 4 * cq, p + 8 * cq, x.data_ptr(), gamma.data_ptr(), y.data_ptr(), A.data_ptr(),
                                       B, C, cq, H, W, bs, ct, bs, ct, bs, ct, H * W * C, C, H * W * C, C, ws.data_ptr(), nf, st))
    L.check(L.ccnet_cca_backward_pm_f32(dy.data_ptr(), p, p + 4 * cq, p + 8 * cq, A.data_ptr(), gamma.data_ptr(), gq, gq + 4 * cq,
                                        gq + 8 * cq, dg.data_ptr(), scr.data_ptr(), B, C, cq, H, W, H * W * C, C, bs, ct, bs, ct,
                                        bs, ct, bs, ct, bs, ct, bs, ct, ws.data_ptr(), nb, st))


for _ in range(3):
    step()
torch.cuda.synchronize()
L.set_option("planes_overlap", 0)
rec = L.profile_launches(lambda: [step() for _ in range(5)])
L.set_option("planes_overlap", -1)
n = len(rec) // 5
print(f"all-pixel-major fp32 core ({B},{C},{H},{W}): {n} launches, event sum {sum(t for _, t in rec) / 5:.4f} ms")
for i in range(n):
    print(f"  {sum(rec[r * n + i][1] for r in range(5)) / 5 * 1e3:8.1f} us  {rec[i][0][:110]}")
