"""Test-only stand-in workload for bench.py's host logic on the gloo backend: the device op has no CPU path, so the
CPU oracle does the arithmetic of one step here.  Used via ``--workload-factory bench_standin:factory``; the product
(bench.py's default path) never imports this module."""
import torch

from oracle import cca_oracle as O


class OracleWorkload:
    def __init__(self, B, C, H, W, device, seed):
        g = torch.Generator().manual_seed(seed)
        Cq = max(C // 8, 1)
        self.q, self.k = torch.randn(B, Cq, H, W, generator=g), torch.randn(B, Cq, H, W, generator=g)
        self.v, self.x, self.dy = (torch.randn(B, C, H, W, generator=g) for _ in range(3))
        self.gamma = torch.full((1,), 0.5)
        self.steps = 0

    def step(self):
        y, A = O.cca_core_forward(self.q, self.k, self.v, self.x, self.gamma)
        O.cca_core_backward(self.dy, self.q, self.k, self.v, A, self.gamma)
        self.steps += 1


def factory(B, C, H, W, device, seed):
    return OracleWorkload(B, C, H, W, device, seed)
