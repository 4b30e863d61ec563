#!/usr/bin/env python3
"""A/B of library options on the headline split-plane step, on ONE set of buffers inside ONE process (boxes of the pool differ
by 3-4 %, so only same-run comparisons count).  Per option set: eager step ms, hipGraph replay ms, fwd / bwd ms and -- with
every launch on one stream -- the per-launch durations.
usage: ab_options.py [B C H W] [--sets name=opt:val,opt:val ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ccnet_amd import _lib  # noqa: E402

args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and sys.argv[i] != "--sets"]
B, C, H, W = (int(a) for a in args[:4]) if len(args) >= 4 else (8, 512, 97, 97)
DEFAULT_SETS = {
    "default": {},
    "v-as-planes": {"_direct": 0},           # the forward splits v into planes first (what maps beyond 100 positions run)
    "no-xcd": {"planes_xcd": 0},
    "no-energy-tail": {"energy_tail": 0},
    "dqdk-2-per-cu": {"dqdk_wpc3": 0},       # ca_backward on the two-slot / two-workgroups-per-CU form
    "dA-3-stages": {"da_stages": 3},         # the persistent dA kernel with three ring stages: it fills the LDS, nothing runs next to it
    "dA-3-stages-1s": {"da_stages": 3, "planes_overlap": 0},    # the energies launch as one workgroup per strip (2.02 rounds -> three)
    "one-stream": {"planes_overlap": 0},
    "overlap-1": {"planes_overlap": 1},
    "dqdk-three-terms": {"dqdk_exact": 0},   # ca_backward with three bf16 terms per product instead of six (round 6 default: six, fp32-equivalent)
    "ring-2-per-cu": {"planes_ring": 1},     # the column ring passes with three slots and TWO workgroups per CU (slow-box A/B)
}
if "--sets" in sys.argv:
    keep = sys.argv[sys.argv.index("--sets") + 1].split(",")
    DEFAULT_SETS = {k: v for k, v in DEFAULT_SETS.items() if k in keep or k == "default"}
DETAIL = ("one-stream", "dqdk-three-terms")
lib = _lib.get_lib()
dev = torch.device("cuda:0")
BASE = {"planes_ring": 2, "planes_stream": 1, "planes_overlap": -1, "planes_xcd": 1, "energy_tail": 1, "da_stages": 2, "dqdk_wpc3": 1,
        "dqdk_exact": 1}
wl = bench.PlanesWorkload(lib, B, C, H, W, dev, 1234)
ref = None
for rnd in range(2):                       # two rounds: the order of the sets must not matter
    for name, opts in DEFAULT_SETS.items():
        for k, v in {**BASE, **opts}.items():
            if not k.startswith("_"):
                lib.set_option(k, v)
        wl.direct = bool(opts.get("_direct", 1)) and max(H, W) <= 100          # (the plane-free form serves strips <= 100)
        if not wl.direct and wl.vpl is None:
            wl.vpl = torch.empty(B, H, W, 2, C, dtype=torch.int16, device=dev)
        for _ in range(5):
            wl.step()
        torch.cuda.synchronize()
        out = (wl.y.clone(), wl.dqkv.clone(), wl.dgamma.clone())
        if ref is None:
            ref = out
        same = all(torch.equal(a, b) for a, b in zip(out[1:], ref[1:])) and float((out[0] - ref[0]).abs().max()) < 1e-5
        ms = bench.time_region(wl.step, 50)
        fwd, bwd = bench.time_region(wl.forward, 30), bench.time_region(wl.backward, 30)
        g = bench.capture_step_graph(wl.step)
        g.replay()
        gms = bench.time_region(g.replay, 50)
        del g
        print(f"== round {rnd} {name:16s} {opts}: eager {ms:.4f} ms  graph {gms:.4f} ms  fwd {fwd:.4f}  bwd {bwd:.4f}  bit-identical to first: {same}", flush=True)
        if rnd == 0 and name in DETAIL:
            lib.set_option("planes_overlap", 0)          # per-launch durations: every launch on one stream
            rec = lib.profile_launches(lambda: [wl.step() for _ in range(5)])
            n = len(rec) // 5
            print(f"     launches {n}, event sum {sum(t for _, t in rec) / 5:.4f} ms")
            for i in range(n):
                print(f"     {sum(rec[r * n + i][1] for r in range(5)) / 5 * 1e3:8.1f} us  {rec[i][0][:120]}")
            lib.set_option("planes_overlap", -1)
for k, v in BASE.items():
    lib.set_option(k, v)
