#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/summary.json -> profiles/traffic_latest.json  (HBM bytes per launch per kernel).

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced streaming read (checked here on softmax_fwd_kernel, which
reads the 58.4 MB attention tensor once: FETCH_SIZE says 29.4 MB), so fetch bytes are doubled.  WRITE_SIZE
matched the known write volume of the same kernel (57.0 MiB vs 58.4 MB) and is used as is."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
names = {  # rocprof kernel name -> bench.py roofline label
    "cca::weight_strip_kernel<8, false, true>": "weight_strip_kernel ca_map_backward.dA[dy.v]",
    "cca::weight_strip_kernel<8, false, false>": "weight_strip_kernel ca_map_backward.dA[dy.v]",
    "cca::weight_strip_kernel<8, true, false>": "weight_strip_kernel ca_forward[q.k]",
    "cca::map_strip_kernel<8, true, false, 1, true>": "map_strip_kernel<row> ca_map_forward[A.v]",
    "cca::map_strip_kernel<8, false, true, 0, false>": "map_strip_kernel<col> ca_map_backward.dv[A^T.dy]",
    "cca::map_strip_kernel<8, true, true, 1, true>": "map_strip_kernel<row> ca_map_backward.dv[A^T.dy]",
    "cca::map_strip_kernel<8, false, false, 2, false>": "map_strip_kernel<col> ca_map_forward[A.v]",
    "cca::map_strip_kernel<8, true, false, 1, false>": "map_strip_kernel<row> ca_map_forward[A.v]",
}
d = json.load(open(src))
out = {}
for k, v in d.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        nbytes = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        out[names.get(k, k)] = nbytes
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
