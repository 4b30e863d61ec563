#!/usr/bin/env python3
"""Round 5 probe for the module-level items of VERDICT r4 (5, 8): what do the projection GEMMs cost in the forms the node could use?
 (a) forward: ONE stacked bf16 -> fp32 GEMM (M = B HW, N = 640, K = 1536)  vs  q | k (N = 128) and v (N = 512) as two GEMMs
     (the split lets energies + softmax run next to the v GEMM);
 (b) backward dx: bmm(out_dtype=fp32).add_(dy)  vs  baddbmm(dy, ..., out_dtype=fp32) (beta = 1 in the GEMM).
Timed with HIP events, 20 iterations each, one process."""
import torch

dev = torch.device("cuda:0")
B, C, H, W = 8, 512, 97, 97
cq, hw = C // 8, H * W
ct = 2 * cq + C


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x3 = torch.randn(B * hw, 3 * C, device=dev).to(torch.bfloat16)
w3buf = torch.randn(ct, 3 * C, device=dev).to(torch.bfloat16)
w3 = w3buf.t()
bias = torch.randn(ct, device=dev)
print("fwd one GEMM N=640        %.1f us" % t(lambda: torch.addmm(bias, x3, w3, out_dtype=torch.float32)))
wqk, wv = w3buf[:2 * cq].t(), w3buf[2 * cq:].t()
print("fwd q|k GEMM N=128        %.1f us" % t(lambda: torch.addmm(bias[:2 * cq], x3, wqk, out_dtype=torch.float32)))
print("fwd v GEMM N=512          %.1f us" % t(lambda: torch.addmm(bias[2 * cq:], x3, wv, out_dtype=torch.float32)))
out = torch.empty(B * hw, ct, device=dev)
try:
    f = lambda: (torch.addmm(bias[:2 * cq], x3, wqk, out_dtype=torch.float32, out=out[:, :2 * cq]),      # noqa: E731
                 torch.addmm(bias[2 * cq:], x3, wv, out_dtype=torch.float32, out=out[:, 2 * cq:]))
    print("fwd two GEMMs into slices of one packed tensor (out=)  %.1f us" % t(f))
    ref = torch.addmm(bias, x3, w3, out_dtype=torch.float32)
    print("   max |packed - single| = %.3e" % float((out - ref).abs().max()))
except Exception as e:
    print("fwd out= into slices: not supported:", str(e)[:200])

d3 = torch.randn(B, hw, 3 * ct, device=dev).to(torch.bfloat16)
w3t = torch.randn(C, 3 * ct, device=dev).to(torch.bfloat16)
dy = torch.randn(B, C, hw, device=dev)
a = lambda: torch.bmm(w3t.unsqueeze(0).expand(B, -1, -1), d3.transpose(1, 2), out_dtype=torch.float32).add_(dy)      # noqa: E731
print("bwd dx  bmm + add_        %.1f us" % t(a))
try:
    b = lambda: torch.baddbmm(dy, w3t.unsqueeze(0).expand(B, -1, -1), d3.transpose(1, 2), out_dtype=torch.float32)   # noqa: E731
    print("bwd dx  baddbmm(out_dtype) %.1f us" % t(b))
    print("   max |baddbmm - (bmm + add)| = %.3e" % float((b() - a()).abs().max()))
except Exception as e:
    print("bwd baddbmm(out_dtype): not supported:", str(e)[:200])
print("bwd dx  bmm alone         %.1f us" % t(lambda: torch.bmm(w3t.unsqueeze(0).expand(B, -1, -1), d3.transpose(1, 2), out_dtype=torch.float32)))
x3b = x3.view(B, 3 * hw, C)
print("bwd dW  bmm + sum         %.1f us" % t(lambda: torch.bmm(d3.view(B, 3 * hw, ct).transpose(1, 2), x3b, out_dtype=torch.float32).sum(0)))
dq = torch.randn(B, hw, ct, device=dev)
print("bwd db  sum over pixels   %.1f us" % t(lambda: dq.sum(dim=(0, 1))))
# dW as ONE GEMM over all B * 3 HW rows (K = 226 k, a 640 x 512 output: needs split-K inside the library) instead of B batched ones + a sum
d3f, x3f = d3.view(B * 3 * hw, ct), x3.view(B * 3 * hw, C)
try:
    one = lambda: torch.mm(d3f.t(), x3f, out_dtype=torch.float32)                                   # noqa: E731
    print("bwd dW  one mm over B*3HW rows  %.1f us" % t(one))
    ref = torch.bmm(d3.view(B, 3 * hw, ct).transpose(1, 2), x3b, out_dtype=torch.float32).sum(0)
    print("   max |one - (bmm + sum)| / |ref|max = %.3e" % float((one() - ref).abs().max() / ref.abs().max()))
except Exception as e:
    print("bwd dW one mm: failed:", str(e)[:200])
for nb in (2, 4, 16, 32):
    try:
        d3n, x3n = d3.view(nb, B * 3 * hw // nb, ct) if (B * 3 * hw) % nb == 0 else None, x3.view(nb, B * 3 * hw // nb, C) if (B * 3 * hw) % nb == 0 else None
        if d3n is None:
            continue
        print("bwd dW  bmm over %2d row blocks + sum  %.1f us" % (nb, t(lambda: torch.bmm(d3n.transpose(1, 2), x3n, out_dtype=torch.float32).sum(0))))
    except Exception as e:
        print("bwd dW", nb, "blocks: failed:", str(e)[:120])
