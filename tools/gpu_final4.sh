#!/bin/bash
# Round-4 closing GPU call: the full round (tools/gpu_round4.sh) + PMC passes of the final sources + module timings + the stress run.
# usage (repo root on the GPU box): bash tools/gpu_final4.sh <tag>
set -u
TAG=${1:-r4z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
bash tools/gpu_round4.sh "$TAG"
OUT=$R/gpurun_out/$TAG
echo "== module, small batches (eager vs graphed)"; timeout 600 python tools/module_small_batch.py > "$OUT/module_small_batches.txt" 2>&1; tail -4 "$OUT/module_small_batches.txt"
echo "== PMC passes (fp32 step)"; bash tools/pmc.sh "$TAG" --iters 3 > "$OUT/pmc.log" 2>&1; tail -3 "$OUT/pmc.log"
cp "$R/gpurun_out/pmc_$TAG/summary.json" "$OUT/pmc_step_summary.json" 2>/dev/null
echo "== PMC passes (bf16 configs[4] step)"; bash tools/pmc.sh "${TAG}_bf16" --script tools/pm_bf16_time.py > "$OUT/pmc_bf16.log" 2>&1; tail -2 "$OUT/pmc_bf16.log"
cp "$R/gpurun_out/pmc_${TAG}_bf16/summary.json" "$OUT/bf16_config5_pmc_summary.json" 2>/dev/null
echo "== stress (run-to-run bit identity under HBM load)"; timeout 600 python tools/stress_pm.py 100 > "$OUT/stress_pm.log" 2>&1; tail -3 "$OUT/stress_pm.log"
echo "== kernel-trace timelines (eager / graph-replayed steps: who really overlaps whom)"; bash tools/timeline.sh "$TAG" > "$OUT/timeline.log" 2>&1; tail -3 "$OUT/timeline_graph.txt"
echo "== bench, bf16 configs[4] as the measured workload"; timeout 600 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-train --no-cpu-baseline > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; head -c 600 "$OUT/bench_bf16.json"; echo
# keep the payload small
find "$R/gpurun_out/pmc_$TAG" "$R/gpurun_out/pmc_${TAG}_bf16" -name "*.csv" -size +5M -delete 2>/dev/null
echo "== final done"
