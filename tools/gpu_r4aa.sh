#!/bin/bash
# round 4, call 15: bf16 column partials (A/B + parity), where the B = 1 module step goes
TAG=${1:-r4aa}; R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== bf16 parity tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -m gpu -k "bf16" > $OUT/pytest_bf16.log 2>&1; tail -3 $OUT/pytest_bf16.log
grep -h "excess" $OUT/pytest_bf16.log | tail -14
echo "== bf16 A/B"; timeout 600 python tools/ab_bf16.py > $OUT/ab_bf16.txt 2>&1; cat $OUT/ab_bf16.txt
echo "== module B=1,2: eager / host issue / one graph / graph_module"; timeout 600 python tools/probes/module_b1_hostbound.py > $OUT/module_b1.txt 2>&1; grep "^B=" $OUT/module_b1.txt
echo "== module B=1 kernel sum (rocprofv3, 200 eager steps)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b1 -- python $R/tools/probes/module_b1_hostbound.py --eager-only 200 --batches 1 > $OUT/prof_b1.log 2>&1
cd $R; f=$(find $OUT/prof_b1 -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/module_b1_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/module_b1_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print(f"kernel sum per step: {tot/205/1e3:.1f} us over {calls/205:.1f} launches per step")
for r in rows[:22]: print(f'  {float(r["TotalDurationNs"])/205/1e3:7.1f} us/step  x{int(r["Calls"])/205:.1f}  {r["Name"][:110]}')
PY
find $OUT/prof_b1 -name "*.csv" -size +2M -delete 2>/dev/null
echo "== done"
