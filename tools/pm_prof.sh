#!/bin/bash
# per-kernel durations of one pixel-major fwd+bwd (tools/pm_bf16_time.py) under rocprofv3; usage: bash tools/pm_prof.sh <tag> [B C H W [bf16|f32]]
TAG=${1:-pm}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
timeout 200 python "$R/tools/pm_bf16_time.py" "$@" 2>&1 | tail -1 | tee "$OUT/time.txt"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$R/tools/pm_bf16_time.py" "$@" > "$OUT/out.txt" 2>&1
cd "$R"
T=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/pm_seq.py "$T" | tee "$OUT/seq.txt"
find "$OUT/prof" -name "*kernel_trace*.csv" -size +20M -delete
exit 0
