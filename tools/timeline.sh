#!/bin/bash
# Kernel-trace TIMELINE (start / end stamps, not --stats) of a few bench steps, eager and graph-replayed.
# usage (GPU box, repo root): bash tools/timeline.sh <tag> [bench args...]
TAG=${1:-tl}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$mode" -- python "$R/bench.py" --steps 12 --warmup 4 --prewarm-s 0.2 --no-extras --launch $mode "$@" > "$OUT/trace_${mode}_bench.json" 2> "$OUT/trace_$mode.err"
  echo "trace $mode rc=$?"
  python3 "$R/tools/timeline.py" "$OUT/trace_$mode" 3 > "$OUT/timeline_$mode.txt" 2>&1
  tail -4 "$OUT/timeline_$mode.txt"
  find "$OUT/trace_$mode" -name "*.csv" -size +8M -delete
done
