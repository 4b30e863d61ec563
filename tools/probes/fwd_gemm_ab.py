#!/usr/bin/env python3
"""A/B of the module's forward projection GEMM on its real operands (x as three bf16 planes (B HW, 3 C), the packed weight (3 C, ct) bf16,
fp32 output (B HW, ct) with the bias): one addmm (what ships) against the same product as B batch entries, and without the bias epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 512, 97, 97)
hw, ct = H * W, C + 2 * (C // 8)
torch.manual_seed(0)
x3 = torch.randn(B * hw, 3 * C, device=dev).to(torch.bfloat16)
w3 = torch.randn(ct, 3 * C, device=dev).to(torch.bfloat16).t()          # (3C, ct) view, as _pack_projection hands it out
bias = torch.randn(ct, device=dev)
variants = {
    "addmm(bias, x3, w3, out_dtype=fp32)  [ships]": lambda: torch.addmm(bias, x3, w3, out_dtype=torch.float32),
    "mm(x3, w3, out_dtype=fp32), no bias": lambda: torch.mm(x3, w3, out_dtype=torch.float32),
    "mm + add_(bias)": lambda: torch.mm(x3, w3, out_dtype=torch.float32).add_(bias),
    "baddbmm over B entries": lambda: torch.baddbmm(bias.view(1, 1, -1), x3.view(B, hw, 3 * C), w3.unsqueeze(0).expand(B, -1, -1), out_dtype=torch.float32),
    "bmm over B entries, no bias": lambda: torch.bmm(x3.view(B, hw, 3 * C), w3.unsqueeze(0).expand(B, -1, -1), out_dtype=torch.float32),
    "addmm on a contiguous weight": None,
}
from ccnet_amd import _lib  # noqa: E402
lib = _lib.get_lib()
wt = w3.t()                                                               # (ct, 3C) contiguous: the packed buffer itself
assert wt.is_contiguous()
out_hw = torch.empty(B * hw, ct, device=dev)


def handwritten():
    lib.check(lib.ccnet_cca_projection_bf16(x3.data_ptr(), wt.data_ptr(), bias.data_ptr(), out_hw.data_ptr(), B * hw, ct, 3 * C,
                                            3 * C, 3 * C, ct, torch.cuda.current_stream().cuda_stream), "projection_bf16")
    return out_hw


variants["hand-written MFMA GEMM, bias in the accumulators (csrc/cca_gemm.hpp)"] = handwritten
w3c = w3.contiguous()
variants["addmm on a contiguous weight"] = lambda: torch.addmm(bias, x3, w3c, out_dtype=torch.float32)
ref = None
for rnd in range(2):
    for name, f in variants.items():
        try:
            for _ in range(3):
                r = f()
            torch.cuda.synchronize()
            if ref is None:
                ref = r.view(B * hw, ct).clone()
            err = float((r.view(B * hw, ct) - ref).abs().max())
            print(f"round {rnd}: {name:72s} {bench.time_region(f, 30) * 1e3:8.1f} us   max |diff| vs first {err:.2e}", flush=True)
        except Exception as e:
            print(f"round {rnd}: {name:72s} failed: {e}", flush=True)
