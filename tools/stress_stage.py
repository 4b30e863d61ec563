import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H = int(sys.argv[2]) if len(sys.argv) > 2 else 97
W = int(sys.argv[3]) if len(sys.argv) > 3 else 97
wl = bench.CoreWorkload(lib, 8, 512, H, W, dev, 7)
print('shape', (8, 512, H, W))
noise = torch.randn(64 * 1024 * 1024, device=dev); side = torch.cuda.Stream()
wl.forward(); torch.cuda.synchronize()
stages = dict(wl.stage_table()); 
names = {"dA": "ca_map_backward.dA[dy.v]", "dv": "ca_map_backward.dv[A^T.dy]", "map_fwd": "ca_map_forward[A.v]", "weight_fwd": "ca_forward[q.k]"}
outs = {"dA": lambda: wl.scratch, "dv": lambda: wl.dv, "map_fwd": lambda: wl.y, "weight_fwd": lambda: wl.scratch}
for key, nm in names.items():
    fn = stages[nm][0]
    lib.check(fn(), nm); torch.cuda.synchronize(); ref = outs[key]().clone(); bad = 0
    for i in range(iters):
        if i % 2:
            with torch.cuda.stream(side): noise.mul_(1.0001)
        lib.check(fn(), nm); torch.cuda.synchronize()
        bad += int(not torch.equal(outs[key](), ref))
    print(key, "differing runs:", bad, "/", iters, flush=True)
# dual
dE = torch.randn_like(wl.A)
f = wl.extra_stages()["ca_backward[dq,dk]"]; wl.scratch.copy_(dE)
lib.check(f(), "dual"); torch.cuda.synchronize(); r1, r2 = wl.dq.clone(), wl.dk.clone(); bad = 0
for i in range(iters):
    if i % 2:
        with torch.cuda.stream(side): noise.mul_(1.0001)
    lib.check(f(), "dual"); torch.cuda.synchronize()
    bad += int(not (torch.equal(wl.dq, r1) and torch.equal(wl.dk, r2)))
print("dual dq/dk differing runs:", bad, "/", iters, flush=True)
