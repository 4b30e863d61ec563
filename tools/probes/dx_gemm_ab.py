#!/usr/bin/env python3
"""A/B for VERDICT r5 item 5a: dx = dy + W^T dqkv^T of the module's backward as ``bmm(out_dtype=fp32).add_(dy)`` (rounds 4-5: a
98 us elementwise pass over dx) against ONE ``baddbmm(dy, ..., out_dtype=fp32)`` (dy as the GEMM's C operand, beta = 1), on the
operands the split-plane node really has: W^T as (C, 3 ct) bf16 planes, dqkv as (B, HW, 3 ct) bf16 three-plane rows, dy NCHW fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 512, 97, 97)
hw, ct = H * W, C + 2 * (C // 8)
torch.manual_seed(0)
w3t = torch.randn(C, 3 * ct, device=dev).to(torch.bfloat16)
d3 = torch.randn(B, hw, 3 * ct, device=dev).to(torch.bfloat16)
dy = torch.randn(B, C, hw, device=dev)
A = w3t.unsqueeze(0).expand(B, -1, -1)
Bm = d3.transpose(1, 2)


def two_pass():
    return torch.bmm(A, Bm, out_dtype=torch.float32).add_(dy)


def folded():
    return torch.baddbmm(dy, A, Bm, out_dtype=torch.float32)


def gemm_only():
    return torch.bmm(A, Bm, out_dtype=torch.float32)


def library():
    """ccnet_cca_projection_adjoint_bf16: the library's own GEMM, dy starting the accumulators (csrc/cca_gemm.hpp)"""
    from ccnet_amd import functions as F, _lib
    return F._projection_adjoint_gemm(_lib.get_lib(), w3t, d3, dy)


def library_no_add():
    from ccnet_amd import _lib
    lib = _lib.get_lib()
    out = torch.empty((B, C, hw), device=dev)
    lib.check(lib.ccnet_cca_projection_adjoint_bf16(w3t.data_ptr(), d3.data_ptr(), None, out.data_ptr(), B, C, hw, 3 * ct, 3 * ct, 3 * ct,
                                                    hw * 3 * ct, torch.cuda.current_stream().cuda_stream))
    return out


for f in (two_pass, folded, gemm_only, library, library_no_add):
    for _ in range(3):
        f()
torch.cuda.synchronize()
r0, r1 = two_pass(), folded()
print("max |bmm + add - baddbmm| =", float((r0 - r1).abs().max()), " (|dx|max", float(r0.abs().max()), ")")
print("max |bmm + add - library GEMM| =", float((r0 - library()).abs().max()))
for rnd in range(2):
    for name, f in (("bmm(out_dtype=fp32).add_(dy)", two_pass), ("baddbmm(dy, ..., out_dtype=fp32)", folded), ("bmm alone", gemm_only),
                    ("library GEMM, dy in the accumulators", library), ("library GEMM alone", library_no_add)):
        print(f"round {rnd}: {name:36s} {bench.time_region(f, 30) * 1e3:8.1f} us", flush=True)

# ---- the weight gradient: dW = sum over images of dqkv_planes^T . x_planes (K = 3 HW rows per image, M = ct, N = C): 20 output tiles
# ---- of 128 x 128 per batch entry -- 8 entries leave most CUs idle.  The K axis (rows: pixel-major, three planes per pixel) splits
# ---- into any number of contiguous row ranges: s x more batch entries of K / s rows each, summed by the same .sum(0)
x3 = torch.randn(B, 3 * hw, C, device=dev).to(torch.bfloat16)
ref = None
for s in (1, 3, 97):
    if (3 * hw) % s:
        continue
    a = d3.view(B * s, 3 * hw // s, ct).transpose(1, 2)
    b = x3.view(B * s, 3 * hw // s, C)

    def dw():
        return torch.bmm(a, b, out_dtype=torch.float32).sum(0)

    for _ in range(3):
        r = dw()
    torch.cuda.synchronize()
    if ref is None:
        ref = r
    print(f"dW as {B * s:4d} batch entries of K = {3 * hw // s:6d}: {bench.time_region(dw, 20) * 1e3:8.1f} us   max |diff| vs s = 1: "
          f"{float((r - ref).abs().max()):.2e} (|dW|max {float(ref.abs().max()):.1f})", flush=True)

# ---- the library's row-contraction GEMM (ccnet_cca_projection_wgrad_bf16, csrc/cca_gemm.hpp): S slabs x 10 output tiles
from ccnet_amd import functions as F, _lib  # noqa: E402
lib = _lib.get_lib()
d2, x2 = d3.view(B * hw * 3, ct), x3.view(B * hw * 3, C)
want = torch.bmm(d3.view(B, 3 * hw, ct).transpose(1, 2), x3.view(B, 3 * hw, C), out_dtype=torch.float32).sum(0)
for _ in range(3):
    got = F._projection_wgrad_gemm(lib, d2, x2)
torch.cuda.synchronize()
print(f"dW by the library's GEMM + sum of partials:  {bench.time_region(lambda: F._projection_wgrad_gemm(lib, d2, x2), 30) * 1e3:8.1f} us"
      f"   max |diff| vs s = 1: {float((got - want).abs().max()):.2e}", flush=True)
for S in (12, 25, 51):
    part = torch.empty((S, ct, C), device=dev)
    f = lambda: lib.check(lib.ccnet_cca_projection_wgrad_bf16(d2.data_ptr(), x2.data_ptr(), part.data_ptr(), B * hw * 3, ct, C, ct, C, S,
                                                               torch.cuda.current_stream().cuda_stream))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    print(f"   the launch alone, S = {S:3d} ({S * 10} workgroups): {bench.time_region(f, 30) * 1e3:8.1f} us", flush=True)

