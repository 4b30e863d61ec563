import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccnet_amd import CrissCrossAttention
dev = torch.device("cuda:0")
m = CrissCrossAttention(512).to(dev)
with torch.no_grad():
    m.gamma.fill_(0.5)
for shape in [(1, 512, 33, 18), (1, 512, 97, 97), (2, 512, 97, 97)]:
    x = torch.randn(*shape, device=dev, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xb = torch.nn.functional.relu(torch.nn.functional.conv2d(x, torch.randn(512, 512, 1, 1, device=dev) * 0.05))
        print("x into module:", xb.dtype, xb.shape, xb.stride(), flush=True)
        y = m(xb)
        print("y:", y.dtype, y.shape, y.stride(), flush=True)
        z = m(y)
        loss = z.float().square().mean()
    loss.backward()
    torch.cuda.synchronize()
    print("ok", shape, float(loss), flush=True)
