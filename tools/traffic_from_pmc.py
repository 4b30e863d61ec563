#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/summary.json -> profiles/traffic_latest.json  (HBM bytes per launch per kernel).

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced streaming read (checked here on softmax_fwd_kernel, which
reads the 58.4 MB attention tensor once: FETCH_SIZE says 29.4 MB), so fetch bytes are doubled.  WRITE_SIZE
matched the known write volume of the same kernel (57.0 MiB vs 58.4 MB) and is used as is.

The file also records the sha256 prefix of the kernel sources the profiled library was built from (``_src_sha16``: bench.py
only quotes the traffic when it benches a build of those same sources) and the per-step total (every kernel is launched once per step).
``--per-dispatch``: a kernel that a step launches more than once (the split-plane step launches gmap3_kernel<.., 2, 3> for the
forward and the dv column pass) is accounted per dispatch: the per-step total is sum(average * dispatches) / steps.
With ``--steps N`` the profiled command ran N steps in which a kernel may be launched several times (the pixel-major bf16
step, tools/pm_bf16_time.py: 3 warm-up + 10 timed forward/backward pairs = 13): the per-step total is then
sum(kernel average * dispatches) / N, and ``--out`` names the file (profiles/traffic_bf16_latest.json).
The source hash is the one tools/pmc.sh recorded INSIDE the summary at profiling time; a summary without it, or whose hash
differs from the working tree's kernel sources, is refused (re-profile instead of restamping).
usage: traffic_from_pmc.py <summary.json> [--steps N] [--out NAME.json]"""
import argparse, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("--steps", type=int, default=0)
ap.add_argument("--out", default="traffic_latest.json")
ap.add_argument("--exclude", default="", help="comma-separated kernel-name fragments left out of the per-step total (e.g. the "
                                              "producer-side pm_split_kernel of the split-plane workload, which runs outside the step)")
a = ap.parse_args()
src = a.src


def label(k):
    """rocprof kernel name -> bench.py roofline label (None: keep the kernel name)"""
    m = re.match(r"cca::weight_strip_kernel<8, (true|false), (true|false)", k)
    if m:
        return "weight_strip_kernel " + ("ca_forward[q.k]" if m.group(1) == "true" else "ca_map_backward.dA[dy.v]")
    m = re.match(r"cca::map_strip_kernel<8, (true|false), (true|false), (\d)", k)
    if m:
        row, trans = m.group(1) == "true", m.group(2) == "true"
        return f"map_strip_kernel<{'row' if row else 'col'}> " + ("ca_map_backward.dv[A^T.dy]" if trans else "ca_map_forward[A.v]")
    return None


d = json.load(open(src))
sys.path.insert(0, ROOT)
from ccnet_amd import _lib as _cl
sha = d.pop("_src_sha16", None)
if sha is None:
    sys.exit(f"{src}: no _src_sha16 recorded at profiling time (tools/pmc.sh writes it); refusing to stamp it now")
if sha != _cl.kernel_source_sha16():
    sys.exit(f"{src}: profiled sources {sha} != working tree {_cl.kernel_source_sha16()}; re-profile")
out, total = {}, 0
skip = [e for e in a.exclude.split(",") if e]
for k, v in d.items():
    if any(e in k for e in skip):
        continue
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        nbytes = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        out[label(k) or k] = nbytes
        total += nbytes * v.get("dispatches", 1) / a.steps if a.steps else nbytes
out["_step_total_bytes"] = int(total)
out["_src_sha16"] = sha                                # recorded by tools/pmc.sh when the counters were taken
json.dump(out, open(os.path.join(ROOT, "profiles", a.out), "w"), indent=1)
print(json.dumps(out, indent=1))
