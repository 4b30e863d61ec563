"""Repeat the fused core fwd+bwd at the headline shape and count runs whose outputs differ from the first run
(bit-wise) -- a race detector for the counted-vmcnt pipelines.  usage: stress_repeat.py [iters] [B]"""
import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = bench.CoreWorkload(lib, B, 512, 97, 97, dev, 7)
noise = torch.randn(64 * 1024 * 1024, device=dev)       # concurrent HBM pressure from another stream
side = torch.cuda.Stream()
wl.step(); torch.cuda.synchronize()
ref = [t.clone() for t in (wl.y, wl.dq, wl.dk, wl.dv, wl.dgamma)]
bad = {n: 0 for n in ("y", "dq", "dk", "dv", "dgamma")}
for i in range(iters):
    if i % 2:
        with torch.cuda.stream(side):
            noise.mul_(1.0001)
    wl.step()
    torch.cuda.synchronize()
    for n, a, b in zip(bad, (wl.y, wl.dq, wl.dk, wl.dv, wl.dgamma), ref):
        if not torch.equal(a, b):
            bad[n] += 1
print("iters", iters, "B", B, "runs that differ from run 0:", bad, flush=True)
