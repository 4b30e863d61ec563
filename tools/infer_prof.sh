#!/bin/bash
# rocprofv3 kernel stats of the module's INFERENCE forward (no_grad) at a shape -- what the 0.5 ms at (1,512,129,257) are made of.
# usage (GPU box, repo root): bash tools/infer_prof.sh <tag> [B C H W]
TAG=${1:-inferprof}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$R/tools/infer_shape.py" ${@:-1 512 129 257} 40 > "$OUT/infer_shape.txt" 2> "$OUT/prof.err"
cd "$R"
F=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY' | tee "$OUT/infer_kernels.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("calls   avg_us   total_ms  kernel")
for r in rows[:30]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} {float(r['TotalDurationNs'])/1e6:9.2f}  {r['Name'][:140]}")
PY
find "$OUT/prof" -name "*kernel_trace*.csv" -size +20M -delete
cat "$OUT/infer_shape.txt"
