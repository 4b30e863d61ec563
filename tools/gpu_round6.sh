#!/bin/bash
# Round-6 GPU call.  usage (repo root on the GPU box): bash tools/gpu_round6.sh <tag> [ab|tests|bench|prof|pmc ...]
# Everything lands in gpurun_out/<tag>.  Stages run in the order given; default: ab tests bench.
set -u
TAG=${1:-r6}; shift || true
STAGES=${*:-ab tests bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== host: $(nproc) cores; $(lscpu | grep 'Model name' | sed 's/.*: *//')" | tee "$OUT/host.txt"
for st in $STAGES; do
case $st in
smoke)
  echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"; tail -3 "$OUT/smoke.log";;
ab)
  echo "== ab_options"; timeout 900 python tools/ab_options.py ${AB_ARGS:-} > "$OUT/ab_options.txt" 2>&1; echo "ab rc=$?"; grep "==" "$OUT/ab_options.txt"; grep " us " "$OUT/ab_options.txt" | head -80;;
tests)
  echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q -rA --durations=25 --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
  grep -E "^(FAILED|ERROR)|passed|failed|headline max-abs|scale sweep|vs the ORACLE|Error|error" "$OUT/pytest_gpu.log" | tail -70;;
bench)
  echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 50 --warmup 10 ${BENCH_ARGS:---no-train} > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -3 "$OUT/bench.err"
  python3 - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("ms_per_step", "value", "fwd_ms", "bwd_ms", "eager_ms_per_step", "graph_ms_per_step", "gpu_kernel_sum_ms",
                                  "overlap_gain_ms", "module_ms_per_step", "strips_family", "launch_ms", "per_rank", "collective_library")}
    keep["bf16_config5"] = (d.get("bf16_config5") or {}).get("ms_per_step") if isinstance(d.get("bf16_config5"), dict) else d.get("bf16_config5")
    keep["cpu_baseline"] = {k: (d.get("cpu_baseline") or {}).get(k) for k in ("value", "cores_used", "cores_total", "kind", "full_batch_configs1")}
    print(json.dumps(keep, indent=1)[:6000])
except Exception as e:
    print("bench line unreadable:", e)
PY
  ;;
bf16)
  echo "== bench, bf16 configs[4]"; timeout 600 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-train --no-cpu-baseline > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; head -c 700 "$OUT/bench_bf16.json"; echo;;
prof)
  echo "== rocprofv3 kernel stats"
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rocprof rc=$?"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_single" -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras --overlap 0 > "$OUT/prof_single_bench.json" 2> "$OUT/prof_single.err"; echo "rocprof (single stream) rc=$?"
  cd "$R"
  F=$(find "$OUT/prof_single" -name "*kernel_stats*.csv" | head -1); [ -n "$F" ] && head -20 "$F"
  find "$OUT/prof" "$OUT/prof_single" -name "*kernel_trace*.csv" -size +20M -delete;;
pmc)
  echo "== PMC passes (fp32 step)"; bash tools/pmc.sh "$TAG" --iters 3 > "$OUT/pmc.log" 2>&1; tail -3 "$OUT/pmc.log"
  cp "$R/gpurun_out/pmc_$TAG/summary.json" "$OUT/pmc_step_summary.json" 2>/dev/null
  find "$R/gpurun_out/pmc_$TAG" -name "*.csv" -size +5M -delete 2>/dev/null;;
pmcbf16)
  echo "== PMC passes (bf16 configs[4] step)"; bash tools/pmc.sh "${TAG}_bf16" --script tools/pm_bf16_time.py > "$OUT/pmc_bf16.log" 2>&1; tail -2 "$OUT/pmc_bf16.log"
  cp "$R/gpurun_out/pmc_${TAG}_bf16/summary.json" "$OUT/bf16_config5_pmc_summary.json" 2>/dev/null
  find "$R/gpurun_out/pmc_${TAG}_bf16" -name "*.csv" -size +5M -delete 2>/dev/null;;
fwdab)
  echo "== forward projection GEMM A/B"; timeout 300 python tools/probes/fwd_gemm_ab.py > "$OUT/fwd_gemm_ab.txt" 2>&1; cat "$OUT/fwd_gemm_ab.txt";;
dxab)
  echo "== dx GEMM A/B"; timeout 300 python tools/probes/dx_gemm_ab.py > "$OUT/dx_gemm_ab.txt" 2>&1; cat "$OUT/dx_gemm_ab.txt";;
xprobe)
  echo "== XCD exchange probe"; timeout 300 tools/probes/xcd_exchange_probe 16 20 > "$OUT/xcd_exchange_probe.txt" 2>&1; cat "$OUT/xcd_exchange_probe.txt"
  cd /tmp && export TMPDIR=/tmp
  for grp in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/xprobe_$grp" -- "$R/tools/probes/xcd_exchange_probe" 16 3 > "$OUT/xprobe_$grp.log" 2>&1; echo "pmc $grp rc=$?"
  done
  cd "$R"
  python3 - "$OUT" <<'PY' | tee -a "$OUT/xcd_exchange_probe.txt"
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/xprobe_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    if "k<" not in k: continue
    f, w = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    print(f"PMC per launch {k}: fetch 2 x {sum(f) / len(f) / 1024:.1f} MiB (doubled as in tools/traffic_from_pmc.py), write {sum(w) / len(w) / 1024:.1f} MiB")
PY
  ;;
module)
  echo "== module, small batches"; timeout 600 python tools/module_small_batch.py > "$OUT/module_small_batches.txt" 2>&1; tail -6 "$OUT/module_small_batches.txt";;
stress)
  echo "== stress"; timeout 600 python tools/stress_pm.py 100 > "$OUT/stress_pm.log" 2>&1; tail -3 "$OUT/stress_pm.log";;
*) echo "unknown stage $st";;
esac
done
echo "== done"
