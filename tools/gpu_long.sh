#!/bin/bash
# GPU check of the blocked long-COLUMN passes (maps with both sides beyond 132 positions): parity tests + inference timings.
# usage (repo root on the GPU box): bash tools/gpu_long.sh <tag>
set -u
TAG=${1:-r4long}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
echo "== parity (long strips)"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_rows or tall" -s > "$OUT/pytest_long.log" 2>&1; tail -5 "$OUT/pytest_long.log"
echo "== inference timings"
for s in "1 512 161 321" "1 512 193 385" "1 512 257 513" "1 512 257 129" "1 512 129 257"; do
  timeout 300 python tools/infer_shape.py $s 10 >> "$OUT/inference_long_strips.txt" 2>> "$OUT/infer.err"
done
cat "$OUT/inference_long_strips.txt"
echo "== done"
