#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> a TIMELINE of the last few steps of bench.py: per launch its start offset inside the
step, its duration and the GAP since the previous launch ended; per step the span, the sum of the launches' durations and
the idle time between them (a step's wall time minus the kernels it contains).
usage: timeline.py <rocprof output dir> [steps=3]"""
import csv
import glob
import sys

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = glob.glob(d + "/**/*kernel_trace*.csv", recursive=True)
if not files:
    sys.exit("no kernel_trace csv under " + d)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "cca::" not in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.replace("void ", "").split("(")[0]))
rows.sort()
if not rows:
    sys.exit("no cca:: kernels in the trace")
# a step starts at the first kernel that is launched once per step: take the most common launch count among the kernel
# names (once-per-step kernels outnumber the ones a step launches twice; set-up kernels run once in the whole trace)
import collections
counts = collections.Counter(r[2] for r in rows)
per_step = collections.Counter(counts.values()).most_common(1)[0][0]
first = next(r[2] for r in rows if counts[r[2]] == per_step)
starts = [i for i, r in enumerate(rows) if r[2] == first]
# keep only complete steps with the most common length
lens = [b - a for a, b in zip(starts, starts[1:])]
if not lens:
    sys.exit("fewer than two steps in the trace")
n = max(set(lens), key=lens.count)
steps = [rows[a:a + n] for a, b in zip(starts, starts[1:]) if b - a == n]
print(f"{len(rows)} cca launches, {len(steps)} complete steps of {n} launches; showing the last {min(nsteps, len(steps))}")


def busy(st):
    """time at least one launch of the step is running (launches of the library's side stream overlap the main chain)"""
    total, cur_s, cur_e = 0, None, None
    for s, e, _ in st:                      # sorted by start
        if cur_e is None or s > cur_e:
            total += (cur_e - cur_s) if cur_e is not None else 0
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    return total + ((cur_e - cur_s) if cur_e is not None else 0)


tot = []
for si, st in enumerate(steps):
    t0, t1 = st[0][0], max(e for _, e, _ in st)
    tot.append(((t1 - t0) / 1e3, sum(e - s for s, e, _ in st) / 1e3, busy(st) / 1e3))
for st in steps[-nsteps:]:
    t0 = st[0][0]
    prev_end = None
    print(f"--- step: span {(max(e for _, e, _ in st) - t0) / 1e3:8.1f} us   busy (>= 1 launch running) {busy(st) / 1e3:8.1f} us   "
          f"sum of durations {sum(e - s for s, e, _ in st) / 1e3:8.1f} us")
    for s, e, name in st:
        # gap = idle since every earlier launch ended; "ovl" = it started while an earlier launch was still running
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        tag = f"gap {gap:6.1f} us" if gap >= 0 else f"ovl {-gap:6.1f} us"
        print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  {tag}  {name[:96]}")
        prev_end = e if prev_end is None else max(prev_end, e)
spans = sorted(t[0] for t in tot)
sums = sorted(t[1] for t in tot)
idle = sorted(t[0] - t[2] for t in tot)
mid = len(tot) // 2
print(f"=== over {len(tot)} steps: median span {spans[mid]:.1f} us, median sum of launch durations {sums[mid]:.1f} us (concurrent "
      f"launches count twice), median idle inside a step (no launch running) {idle[mid]:.1f} us")
# step-to-step period (includes the host's gap between steps)
per = sorted((b[0][0] - a[0][0]) / 1e3 for a, b in zip(steps, steps[1:]))
if per:
    print(f"=== step period (start to start): median {per[len(per) // 2]:.1f} us, min {per[0]:.1f}, max {per[-1]:.1f}")
