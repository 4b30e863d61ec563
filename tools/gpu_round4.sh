#!/bin/bash
# Round-4 GPU call: smoke + GPU tests + bench + option A/B + rocprofv3 kernel stats (both stream modes).  Everything lands in gpurun_out/<tag>.
# usage (repo root on the GPU box): bash tools/gpu_round4.sh <tag> [quick]
set -u
TAG=${1:-r4}; MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== host: $(nproc) cores; $(lscpu | grep 'Model name' | sed 's/.*: *//')" | tee "$OUT/host.txt"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"; tail -4 "$OUT/smoke.log"
echo "== ab_options"; timeout 600 python tools/ab_options.py > "$OUT/ab_options.txt" 2>&1; echo "ab rc=$?"; grep "==" "$OUT/ab_options.txt"; grep " us " "$OUT/ab_options.txt" | head -30
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|headline max-abs|logit-scale|vs the ORACLE|Error|error" "$OUT/pytest_gpu.log" | tail -40
if [ "$MODE" = "quick" ]; then
  echo "== bench (no extras)"; timeout 600 python bench.py --gpus 1 --steps 50 --warmup 10 --no-train --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; head -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
  exit 0
fi
echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 50 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; head -c 6000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rocprof rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_single" -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras --overlap 0 > "$OUT/prof_single_bench.json" 2> "$OUT/prof_single.err"; echo "rocprof (single stream) rc=$?"
cd "$R"
F=$(find "$OUT/prof_single" -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && head -20 "$F"
find "$OUT/prof" "$OUT/prof_single" -name "*kernel_trace*.csv" -size +20M -delete
echo "== done"
