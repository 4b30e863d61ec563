#!/usr/bin/env python3
"""Time every strip kernel of several builds of the library (ablation / A-B variants) in ONE process.
usage: python tools/ablate.py name=path.so[:precision][,ENV=value...] ...   (first is the baseline)
The environment knobs of a variant are set around every use of its library (the library reads them lazily, once)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ccnet_amd import _lib

def main():
    dev = torch.device("cuda:0")
    B, C, H, W = 8, 512, 97, 97
    res = {}
    rounds = int(os.environ.get("ABL_ROUNDS", "3"))
    libs = []
    envs = {}
    for spec in sys.argv[1:]:
        name, rest = spec.split("=", 1)
        parts = rest.split(",")
        path, prec = (parts[0].split(":") + [None])[:2]
        lib = _lib.CcaLibrary(os.path.join(ROOT, path))
        if prec is not None:
            lib.ccnet_cca_set_precision(int(prec))
        envs[name] = dict(kv.split("=") for kv in parts[1:])
        libs.append((name, lib))

    def use(name):
        for k in [k for e in envs.values() for k in e]:
            os.environ.pop(k, None)
        os.environ.update(envs[name])

    wls = {name: bench.CoreWorkload(lib, B, C, H, W, dev, 1234) for name, lib in libs}
    for name, lib in libs:
        use(name)
        wls[name].step()             # A must be a valid attention for later stages; initialises the lazy knobs
    torch.cuda.synchronize()
    for r in range(rounds):
        for name, lib in libs:
            use(name)
            wl = wls[name]
            _, rows = bench.roofline_object(wl, iters=10)
            step = bench.time_region(wl.step, 10)
            d = res.setdefault(name, {})
            for row in rows:
                d.setdefault(row["kernel"], []).append(row["ms"])
            for k, v in bench.roofline_object.stages.items():
                d.setdefault("STAGE " + k, []).append(v)
            d.setdefault("STAGE forward", []).append(bench.time_region(wl.forward, 10))
            d.setdefault("STAGE backward", []).append(bench.time_region(wl.backward, 10))
            d.setdefault("STEP fwd+bwd", []).append(step)
    names = [n for n, _ in libs]
    keys = list(res[names[0]].keys())
    print("%-62s" % "kernel (min ms over rounds)", " ".join("%12s" % n for n in names))
    for k in keys:
        print("%-62s" % k[:62], " ".join("%12.4f" % min(res[n][k]) for n in names))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_%s.json" % os.environ.get("ABL_TAG", "x")), "w"))

if __name__ == "__main__":
    main()
