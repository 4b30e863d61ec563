#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/summary.json -> profiles/traffic_latest.json  (HBM bytes per launch per kernel).

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced streaming read (checked here on softmax_fwd_kernel, which
reads the 58.4 MB attention tensor once: FETCH_SIZE says 29.4 MB), so fetch bytes are doubled.  WRITE_SIZE
matched the known write volume of the same kernel (57.0 MiB vs 58.4 MB) and is used as is.

The file also records the sha256 prefix of the library the counters were taken on (``_lib_sha16``: bench.py only
quotes the traffic when it benches that same build) and the per-step total (every kernel is launched once per step).
usage: traffic_from_pmc.py <summary.json> [<libccnet_cca.so>]"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "ccnet_amd", "csrc", "libccnet_cca.so")


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
out, total = {}, 0
for k, v in d.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        nbytes = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        out[label(k) or k] = nbytes
        total += nbytes
out["_step_total_bytes"] = total
out["_lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
