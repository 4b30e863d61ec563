"""Probe 2: the module's three projection GEMMs in the orientations CrissCrossPlanesModuleFunction uses, fp32 (today) vs
split-bf16 x3 through bf16 -> fp32 hipBLASLt GEMMs on K-concatenated operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
B, C, H, W = 8, 512, 97, 97
hw, ct = H * W, 640
torch.manual_seed(0)
x = torch.randn(B, C, hw, device=dev)
dy = torch.randn(B, C, hw, device=dev)
w = torch.randn(ct, C, device=dev) * 0.05
bias = torch.randn(ct, device=dev)
dq = torch.randn(B, hw, ct, device=dev)
def T(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    return round(bench.time_region(f, n) * 1e3, 1)

def split(t):
    hi = t.bfloat16()
    return hi, (t - hi.float()).bfloat16()

# ---- forward: qkv (B, hw, ct) = x^T W^T + b
f32 = lambda: torch.baddbmm(bias.view(1, 1, -1), x.transpose(1, 2), w.t().unsqueeze(0).expand(B, -1, -1))
print("fwd fp32 baddbmm us:", T(f32))
xp = x.transpose(1, 2).contiguous()
xh, xl = split(xp)
X3 = torch.cat([xh, xl, xh], dim=2).contiguous()            # (B, hw, 3C)
wh, wl = split(w)
W3 = torch.cat([wh, wh, wl], dim=1).contiguous()             # (ct, 3C)
ref = f32().double()
for name, fn in (("mm + bias add", lambda: torch.mm(X3.view(B * hw, 3 * C), W3.t(), out_dtype=torch.float32).add_(bias)),
                 ("addmm(bias)", lambda: torch.addmm(bias, X3.view(B * hw, 3 * C), W3.t(), out_dtype=torch.float32))):
    try:
        y = fn().view(B, hw, ct)
        print(f"fwd split-bf16 {name}: us {T(fn)}  max |d| vs fp32 {float((y.double() - ref).abs().max()):.2e}")
    except Exception as e:
        print(f"fwd {name} failed: {str(e)[:200]}")

# two GEMMs on the two-plane layout the library's producers already emit ([xh | xl], 2C per pixel): K = 2C, then K = C with beta = 1
X2 = torch.cat([xh, xl], dim=2).contiguous().view(B * hw, 2 * C)
W2 = torch.cat([wh, wh], dim=1).contiguous()
def two_gemms():
    g1 = torch.addmm(bias, X2, W2.t(), out_dtype=torch.float32)
    return torch.addmm(g1, X2[:, :C], wl.t(), out_dtype=torch.float32)
try:
    y = two_gemms().view(B, hw, ct)
    print(f"fwd split-bf16 two GEMMs on [xh|xl]: us {T(two_gemms)}  max |d| vs fp32 {float((y.double() - ref).abs().max()):.2e}")
except Exception as e:
    print(f"fwd two GEMMs failed: {str(e)[:200]}")

# ---- backward dx (B, C, hw) = dy + W^T dqkv^T
g32 = lambda: torch.baddbmm(dy, w.t().unsqueeze(0).expand(B, -1, -1), dq.transpose(1, 2))
print("dx fp32 baddbmm us:", T(g32))
refdx = g32().double()
dh, dl = split(dq)
D3 = torch.cat([dh, dl, dh], dim=2).contiguous()             # (B, hw, 3 ct)
W3t = torch.cat([wh.t(), wh.t(), wl.t()], dim=1).contiguous()    # (C, 3 ct)
cands = {
    "baddbmm(dy, W3t, D3^T)": lambda: torch.baddbmm(dy, W3t.unsqueeze(0).expand(B, -1, -1), D3.transpose(1, 2), out_dtype=torch.float32),
    "bmm(W3t, D3^T) + dy": lambda: torch.bmm(W3t.unsqueeze(0).expand(B, -1, -1), D3.transpose(1, 2), out_dtype=torch.float32).add_(dy),
    "(bmm(D3, W3t^T))^T pixel-major": lambda: torch.bmm(D3, W3t.t().unsqueeze(0).expand(B, -1, -1), out_dtype=torch.float32),
    "mm(D3 flat, W3t^T) pixel-major": lambda: torch.mm(D3.view(B * hw, 3 * ct), W3t.t(), out_dtype=torch.float32),
}
for name, fn in cands.items():
    try:
        r = fn()
        if r.shape == (B, C, hw):
            e = float((r.double() - refdx).abs().max())
        else:
            e = float((r.view(B, hw, C).transpose(1, 2).double() + dy.double() - refdx).abs().max())
        print(f"dx split-bf16 {name}: us {T(fn)}  max |d| {e:.2e}")
    except Exception as e:
        print(f"dx {name} failed: {str(e)[:200]}")

# ---- backward dW (ct, C) = sum_b dqkv_b^T x_b^T
h32 = lambda: torch.bmm(dq.transpose(1, 2), x.transpose(1, 2)).sum(0)
print("dW fp32 bmm+sum us:", T(h32))
h32b = lambda: torch.mm(dq.view(B * hw, ct).t(), xp.view(B * hw, C))
print("dW fp32 one mm (pixel-major x) us:", T(h32b))
refdw = h32().double()
D3f, X3f = D3.view(B * hw, 3 * ct), X3.view(B * hw, 3 * C)
cands = {
    "mm([dh|dl]^T strided, xh strided) + mm(dh^T, xl)": lambda: (lambda r: r[:ct] + r[ct:])(torch.mm(D3f[:, :2 * ct].t(), X3f[:, :C], out_dtype=torch.float32)) + torch.mm(D3f[:, :ct].t(), X3f[:, C:2 * C], out_dtype=torch.float32),
    "mm(D3f^T (3ct x M), xh) rows": lambda: torch.mm(D3f.t(), X3f[:, :C], out_dtype=torch.float32),
}
for name, fn in cands.items():
    try:
        r = fn()
        e = float((r.double() - refdw).abs().max()) if r.shape == (ct, C) else float("nan")
        print(f"dW split-bf16 {name}: us {T(fn)}  max |d| {e:.2e} (max |ref| {float(refdw.abs().max()):.1f})")
    except Exception as e:
        print(f"dW {name} failed: {str(e)[:200]}")
print("splits: x->planes3 via torch us:", T(lambda: torch.cat(list(split(xp)) + [xp.bfloat16()], dim=2)), " dq->D3 via torch us:", T(lambda: torch.cat(list(split(dq)) + [dq.bfloat16()], dim=2)))

# ---- dW as ONE GEMM over 3M rows: X3' = [xh | xh | xl] and D3 = [dh | dl | dh] per pixel, viewed as (3M, C) / (3M, ct) row lists:
# row pairs (dh, xh), (dl, xh), (dh, xl) -- the three products of the split, K = 3 B HW
X3b = torch.cat([xh, xh, xl], dim=2).contiguous().view(3 * B * hw, C)
D3r = D3.view(3 * B * hw, ct)
for name, fn in (("mm(D3r^T, X3b) K = 3M", lambda: torch.mm(D3r.t(), X3b, out_dtype=torch.float32)),
                 ("per image bmm + sum", lambda: torch.bmm(D3.view(B, 3 * hw, ct).transpose(1, 2), X3b.view(B, 3 * hw, C), out_dtype=torch.float32).sum(0)),
                 ("(mm(X3b^T, D3r))^T", lambda: torch.mm(X3b.t(), D3r, out_dtype=torch.float32).t())):
    try:
        r = fn()
        print(f"dW split-bf16 {name}: us {T(fn)}  max |d| {float((r.double() - refdw).abs().max()):.2e} (max |ref| {float(refdw.abs().max()):.1f})")
    except Exception as e:
        print(f"dW {name} failed: {str(e)[:200]}")
W3b = torch.cat([wh, wl, wh], dim=1).contiguous()
fnb = lambda: torch.mm(X3b.view(B * hw, 3 * C), W3b.t(), out_dtype=torch.float32).add_(bias)
print(f"fwd split-bf16 on [xh|xh|xl] x [wh|wl|wh]: us {T(fnb)}  max |d| vs fp32 {float((fnb().view(B, hw, ct).double() - ref).abs().max()):.2e}")
