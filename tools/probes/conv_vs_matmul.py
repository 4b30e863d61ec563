"""Probe: 1x1 Conv2d (MIOpen) vs batched matmul (hipBLASLt) for the q/k/v projections at (8,512,97,97) fp32."""
import torch, time
dev = torch.device("cuda:0")
B, C, H, W = 8, 512, 97, 97
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for O in (512, 64, 640):
    conv = torch.nn.Conv2d(C, O, 1).to(dev)
    dy = torch.randn(B, O, H, W, device=dev)
    def f_conv():
        x.grad = None; conv.zero_grad(set_to_none=True)
        conv(x).backward(dy)
    def f_conv_fwd():
        with torch.no_grad(): conv(x)
    Wm = conv.weight.detach().view(O, C).clone().requires_grad_(True)
    bm = conv.bias.detach().clone().requires_grad_(True)
    def mm():
        x.grad = None; Wm.grad = None; bm.grad = None
        y = torch.matmul(Wm, x.view(B, C, H * W)) + bm.view(1, O, 1)
        y.view(B, O, H, W).backward(dy)
    def mm_fwd():
        with torch.no_grad(): torch.matmul(Wm, x.view(B, C, H * W)) + bm.view(1, O, 1)
    def bad():
        x.grad = None; Wm.grad = None; bm.grad = None
        y = torch.baddbmm(bm.view(1, O, 1), Wm.unsqueeze(0).expand(B, O, C), x.view(B, C, H * W))
        y.view(B, O, H, W).backward(dy)
    print(f"O={O}: conv fwd {timeit(f_conv_fwd):.3f} fwd+bwd {timeit(f_conv):.3f} | matmul fwd {timeit(mm_fwd):.3f} fwd+bwd {timeit(mm):.3f} | baddbmm fwd+bwd {timeit(bad):.3f} ms")
    with torch.no_grad():
        d = (conv(x) - (torch.matmul(Wm, x.view(B, C, H * W)) + bm.view(1, O, 1)).view(B, O, H, W)).abs().max()
    print("   max diff", float(d))
