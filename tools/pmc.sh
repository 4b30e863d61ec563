#!/bin/bash
# PMC passes (separate runs per counter group: TCC slot limits) for a run_stage.py invocation.
# usage: bash tools/pmc.sh <outdir-tag> [--script tools/other.py] <script args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
SCRIPT=tools/run_stage.py
if [ "${1:-}" = "--script" ]; then SCRIPT=$2; shift 2; fi
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -- python "$R/$SCRIPT" "$@" > "$OUT/p$i.log" 2>&1
  echo "pass $i ($grp) rc=$?"
done
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "cca::" not in k: continue
        k = k.split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in sorted(agg.items()):
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]["dispatches"] = max(len(v) for v in d.values())
# the kernel sources the profiled library was built from, hashed NOW (at profiling time, on the box that ran it)
sys.path.insert(0, ".")
from ccnet_amd import _lib as _cl
res["_src_sha16"] = _cl.kernel_source_sha16()
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, d in res.items():
    if k.startswith("_"): continue
    print(k)
    print("   ", {c: (round(v, 1) if v < 1e6 else f"{v:.4g}") for c, v in d.items()})
PY
