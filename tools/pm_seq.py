"""print the last forward and backward kernel sequence of a rocprofv3 kernel trace of tools/pm_bf16_time.py"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void cca::", "").replace("cca::", "").split("(")[0][:64]  # noqa: E731
seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
last_f = max(i for i, s in enumerate(seq) if s[0].startswith("softmax_fwd"))
last_b = max(i for i, s in enumerate(seq) if s[0].startswith("softmax_bwd"))
print("forward:")
for s in seq[last_f - 1:last_f + 3]:
    print("   %-66s %8.1f us" % s)
print("backward:")
for s in seq[last_b - 3:last_b + 6]:
    print("   %-66s %8.1f us" % s)
