#!/bin/bash
# per-kernel durations of the fp32 pixel-major core with NCHW x / y / dy (bench.PixelMajorF32Workload): bash tools/pm_nchw_prof.sh <tag> <B>
TAG=${1:-pmn}; B=${2:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cat > /tmp/pmn_run.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, bench
from ccnet_amd import _lib
lib = _lib.get_lib()
wl = bench.PixelMajorF32Workload(lib, $B, 512, 97, 97, torch.device("cuda:0"), 1)
for _ in range(10): wl.step()
torch.cuda.synchronize()
print("B=$B pm_nchw step ms", bench.time_region(wl.step, 30))
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python /tmp/pmn_run.py > "$OUT/out.txt" 2>&1
cd "$R"; grep "step ms" "$OUT/out.txt"
F=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void cca::", "").split("(")[0][:74]
    print("%-76s calls %5s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find "$OUT/prof" -name "*kernel_trace*.csv" -size +20M -delete
exit 0
