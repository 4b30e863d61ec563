"""Probe: the three K-concatenated split-bf16 projection GEMMs of the module's default node at (8,512,97,97) under torch's TunableOp
(PYTORCH_TUNABLEOP_ENABLED=1: every GEMM shape is timed once over the rocBLAS / hipBLASLt solution lists and the best one cached)
against torch's default heuristic choice.  Same process: tunable off first, then on.  usage: python tools/probes/tunableop_gemm_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
B, C, hw, ct = 8, 512, 97 * 97, 640
M = B * hw
torch.manual_seed(0)
X3 = torch.randn(M, 3 * C, device=dev).bfloat16()                       # [xh | xh | xl] rows
W3 = (torch.randn(3 * C, ct, device=dev) * 0.05).bfloat16()             # (3C, 2Cq + C) view, as the module keeps it (a .t() of (ct, 3C))
W3 = (torch.randn(ct, 3 * C, device=dev) * 0.05).bfloat16().t()
bias = torch.randn(ct, device=dev)
D3 = torch.randn(B, hw, 3 * ct, device=dev).bfloat16()                  # [dh | dl | dh] rows of dqkv
W3t = (torch.randn(C, 3 * ct, device=dev) * 0.05).bfloat16()
X3b = X3.view(B, 3 * hw, C)
dyv = torch.randn(B, C, hw, device=dev)


def fwd():
    return torch.addmm(bias, X3, W3, out_dtype=torch.float32)


def dx():
    return torch.bmm(W3t.unsqueeze(0).expand(B, -1, -1), D3.transpose(1, 2), out_dtype=torch.float32).add_(dyv)


def dw():
    return torch.bmm(D3.view(B, 3 * hw, ct).transpose(1, 2), X3b, out_dtype=torch.float32).sum(0)


def T(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    return round(bench.time_region(f, n) * 1e3, 1)


for mode in ("default", "tunableop"):
    if mode == "tunableop":
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(30)
        torch.cuda.tunable.set_max_tuning_iterations(50)
    print(f"{mode:10s}: projection fwd {T(fwd)} us   dx {T(dx)} us   dW {T(dw)} us", flush=True)
