#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench + rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [pytest-args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== host: $(nproc) cores; $(lscpu | grep 'Model name' | sed 's/.*: *//')" | tee "$OUT/host.txt"
rocm-smi --showproductname 2>/dev/null | head -12 >> "$OUT/host.txt"

echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"; tail -3 "$OUT/smoke.log"

echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider "$@" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
grep -E "^(PASSED|FAILED|ERROR)|passed|failed|headline max-abs" "$OUT/pytest_gpu.log" | tail -60

echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 50 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"

echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rocprof rc=$?"
# the same command with every launch on one stream: the per-kernel durations the bench line's launch_ms / roofline.launches quote
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_single" -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras --overlap 0 > "$OUT/prof_single_bench.json" 2> "$OUT/prof_single.err"; echo "rocprof (single stream) rc=$?"
cd "$R"
find "$OUT/prof" "$OUT/prof_single" -name "*kernel_stats*.csv" | head -3
F=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && head -30 "$F"
# keep the merged-back payload small: drop the raw per-dispatch trace if it is huge
find "$OUT/prof" "$OUT/prof_single" -name "*kernel_trace*.csv" -size +20M -delete
echo "== done"
