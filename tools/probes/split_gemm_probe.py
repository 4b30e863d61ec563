"""Probe (item f1 / VERDICT r2 item 9): the module's projection GEMMs as split-bf16 x3 on the bf16 matrix pipe through stock
hipBLASLt -- bf16 operands, fp32 output (aten::bmm.dtype) -- against the fp32 GEMM torch runs today.
x^T W^T ~= [x_hi | x_lo] [W_hi ; W_hi]^T + x_hi W_lo^T   (x as pixel-major hi | lo planes, the format the kernels consume)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
B, C, H, W = 8, 512, 97, 97
hw, ct = H * W, 640
torch.manual_seed(0)
x = torch.randn(B, C, hw, device=dev)
w = torch.randn(ct, C, device=dev) * 0.05
b = torch.randn(ct, device=dev)

def split(t):
    hi = t.bfloat16()
    lo = (t - hi.float()).bfloat16()
    return hi, lo

ref64 = torch.baddbmm(b.double().view(1, 1, -1), x.double().transpose(1, 2), w.double().t().unsqueeze(0).expand(B, -1, -1))
f32 = lambda: torch.baddbmm(b.view(1, 1, -1), x.transpose(1, 2), w.t().unsqueeze(0).expand(B, -1, -1))
y32 = f32()
print("fp32 GEMM  max err vs fp64:", float((y32.double() - ref64).abs().max()), " ms:", bench.time_region(f32, 20))
xp = x.transpose(1, 2).contiguous()                       # (B, hw, C) pixel-major
xh, xl = split(xp)
planes = torch.cat([xh, xl], dim=2).contiguous()          # (B, hw, 2C) = hi | lo
wh, wl = split(w)
w2 = torch.cat([wh, wh], dim=1).contiguous()              # (ct, 2C)
try:
    def s3():
        y = torch.bmm(planes, w2.t().unsqueeze(0).expand(B, -1, -1), out_dtype=torch.float32)
        y = torch.baddbmm(y, planes[:, :, :C], wl.t().unsqueeze(0).expand(B, -1, -1), out_dtype=torch.float32)
        return y + b
    y3 = s3()
    print("split-bf16 x3 (2 bf16 GEMMs, fp32 out) max err vs fp64:", float((y3.double() - ref64).abs().max()), " ms:", bench.time_region(s3, 20))
    one = lambda: torch.bmm(planes, w2.t().unsqueeze(0).expand(B, -1, -1), out_dtype=torch.float32)
    print("   first GEMM alone (K = 2C):", bench.time_region(one, 20), "ms")
    flat = planes.view(B * hw, 2 * C)
    mm = lambda: torch.mm(flat, w2.t(), out_dtype=torch.float32)
    print("   as one mm (M = B*hw):", bench.time_region(mm, 20), "ms")
    w3 = torch.cat([wh, wh, wl], dim=1).contiguous()     # [hi | lo | hi] x [Wh ; Wh ; Wl]
    p3 = torch.cat([xh, xl, xh], dim=2).contiguous().view(B * hw, 3 * C)
    mm3 = lambda: torch.mm(p3, w3.t(), out_dtype=torch.float32)
    y33 = mm3().view(B, hw, ct) + b
    print("   single K = 3C mm: err", float((y33.double() - ref64).abs().max()), " ms:", bench.time_region(mm3, 20))
except Exception as e:
    print("bf16 -> fp32 GEMM not available:", str(e)[:300])
# backward-style GEMMs: dx^T = dqkv W  (M = hw, K = ct, N = C) and dW = dqkv^T x (M = ct, K = B*hw, N = C)
dq = torch.randn(B, hw, ct, device=dev)
g1 = lambda: torch.bmm(dq, w.unsqueeze(0).expand(B, -1, -1))
print("fp32 dx GEMM ms:", bench.time_region(g1, 20))
g2 = lambda: torch.mm(dq.view(B * hw, ct).t(), xp.view(B * hw, C))
print("fp32 dW GEMM ms:", bench.time_region(g2, 20))
try:
    dh, dl = split(dq)
    d2 = torch.cat([dh, dl], dim=2).contiguous().view(B * hw, 2 * ct)
    wst = torch.cat([wh, wh], dim=0).contiguous()         # (2 ct, C)
    h1 = lambda: torch.mm(d2, wst, out_dtype=torch.float32)
    print("bf16 dx GEMM (K = 2 ct) ms:", bench.time_region(h1, 20))
    h2 = lambda: torch.mm(d2.t(), torch.cat([xh, xh], dim=2).view(B * hw, 2 * C)[:, :C].contiguous(), out_dtype=torch.float32)
    print("bf16 dW GEMM (M = 2 ct, K = B*hw) ms:", bench.time_region(h2, 20))
except Exception as e:
    print("bf16 backward GEMMs:", str(e)[:300])
