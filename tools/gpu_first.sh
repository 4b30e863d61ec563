#!/bin/bash
# round-3 first GPU call: probe + bench (graph / eager / idle accounting) + kernel-trace timelines + GPU tests
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== host: $(nproc) cores; $(lscpu | grep 'Model name' | sed 's/.*: *//')" | tee "$OUT/host.txt"
rocm-smi --showproductname 2>/dev/null | head -12 >> "$OUT/host.txt"
echo "== tr16 probe"; timeout 60 tools/probes/tr16_probe > "$OUT/tr16_probe.log" 2>&1; echo "rc=$?"; tail -4 "$OUT/tr16_probe.log"
echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 50 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== timelines"; bash tools/timeline.sh $TAG
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|max-abs|max excess|golden|module node" "$OUT/pytest_gpu.log" | tail -60
echo "== done"
