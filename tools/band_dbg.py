import sys, torch
sys.path.insert(0, '/root/repo')
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device('cuda'); s = torch.cuda.current_stream().cuda_stream
B, C, H, W = 1, 32, 33, 18
g = torch.Generator().manual_seed(1)
A = torch.rand(B, H, W, H + W, generator=g).to(dev)
v = torch.zeros(B, C, H, W, device=dev); v[:, :, 32, :] = 1.0
x = torch.zeros(B, C, H, W, device=dev); gamma = torch.ones(1, device=dev)
y = torch.full_like(x, float('nan'))
vpm = v.permute(0, 2, 3, 1).contiguous()
lib.check(lib.ccnet_ca_map_forward_pm_f32(A.data_ptr(), vpm.data_ptr(), x.data_ptr(), gamma.data_ptr(), y.data_ptr(), B, C, H, W, H * W * C, C, s))
torch.cuda.synchronize()
want = A[0, :, :, 32].clone()
want[32] += A[0, 32, :, H:].sum(-1)
got = y[0, 0]
d = (got - want)
torch.set_printoptions(precision=3, linewidth=250, sci_mode=False)
print("err by (h,w):"); print(d[:, :8].cpu())
print("got/want at h=0..3,w=0:", got[:4, 0].cpu(), want[:4, 0].cpu())
# which A element did we get?  search
for h in (0, 1, 5):
    val = float(got[h, 0]); hits = (A[0] - val).abs() < 1e-3
    print(h, val, hits.nonzero()[:5].cpu().tolist())
print("channels equal:", bool((y[0] - y[0, 0:1]).abs().max() < 1e-6))
