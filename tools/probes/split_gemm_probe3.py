"""Probe 3: shapes of the K-concatenated split-bf16 projection GEMM (M = B HW = 75272, K = 3C = 1536): one GEMM with N = 640
(q | k | v) against separate v (N = 512) and q | k (N = 128) GEMMs, N padded to 768, and the operand orders hipBLASLt sees."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
B, C, hw, ct = 8, 512, 97 * 97, 640
M, K = B * hw, 3 * C
torch.manual_seed(0)
X3 = torch.randn(M, K, device=dev).bfloat16()
W3 = (torch.randn(ct, K, device=dev) * 0.05).bfloat16()
def T(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    return round(bench.time_region(f, n) * 1e3, 1)
mm = lambda a, b: torch.mm(a, b, out_dtype=torch.float32)
print("N=640  mm(X3, W3^T):", T(lambda: mm(X3, W3.t())))
Wv, Wqk = W3[128:].contiguous(), W3[:128].contiguous()
print("N=512  mm(X3, Wv^T):", T(lambda: mm(X3, Wv.t())), "  N=128 mm(X3, Wqk^T):", T(lambda: mm(X3, Wqk.t())))
W768 = torch.cat([W3, torch.zeros(128, K, device=dev, dtype=torch.bfloat16)], 0)
print("N=768 (zero-padded) mm:", T(lambda: mm(X3, W768.t())))
W3c = W3.t().contiguous()      # (K, N) row-major: "NN"
print("N=640  mm(X3, W3c) [B as (K, N) row-major]:", T(lambda: mm(X3, W3c)))
print("N=640  (mm(W3, X3^T))^T [out (N, M)]:", T(lambda: mm(W3, X3.t())))
out = torch.empty(M, ct, device=dev)
print("N=640  mm(out=) preallocated:", T(lambda: torch.mm(X3, W3.t(), out_dtype=torch.float32, out=out)))
# K = 1920, N = 512 (the dx shape that measured 153 us pixel-major)
D3 = torch.randn(M, 3 * ct, device=dev).bfloat16(); W3t = (torch.randn(C, 3 * ct, device=dev) * 0.05).bfloat16()
print("dx-shaped: M x 1920 x 512 mm(D3, W3t^T):", T(lambda: mm(D3, W3t.t())))
# bf16 output instead of fp32 (to see whether the fp32 store is the cost)
print("N=640 bf16 out:", T(lambda: torch.mm(X3, W3.t())))
