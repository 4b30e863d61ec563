import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from ccnet_amd import _lib
lib = _lib.get_lib(); dev = torch.device("cuda:0")
wl = bench.PlanesWorkload(lib, 8, 512, 97, 97, dev, 1234)
res = {}
for ring in (2, 12, 2, 12):
    lib.set_option("planes_ring", ring)
    for ov in (0, -1):
        lib.set_option("planes_overlap", ov)
        for _ in range(5): wl.step()
        torch.cuda.synchronize()
        ms = bench.time_region(wl.step, 40); bw = bench.time_region(wl.backward, 40)
        print(f"planes_ring={ring} overlap={ov}: step {ms:.4f} bwd {bw:.4f}", flush=True)
    lib.set_option("planes_overlap", 0)
    rec = lib.profile_launches(lambda: [wl.backward() for _ in range(5)])
    n = len(rec) // 5
    print("   ", [(rec[i][0][5:40], round(sum(rec[r*n+i][1] for r in range(5))/5*1e3,1)) for i in range(n)][-5:])
    wl.step(); torch.cuda.synchronize(); res[ring] = wl.dqkv.clone()
print("bit-identical dqkv:", torch.equal(res[2], res[12]))
lib.set_option("planes_ring", 2); lib.set_option("planes_overlap", -1)
