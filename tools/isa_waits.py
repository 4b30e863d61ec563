#!/usr/bin/env python3
"""Summarise the vector-memory wait structure of one kernel in a hipcc --save-temps .s file: every `s_waitcnt vmcnt(N)` that
is NOT part of a counted-barrier ladder (i.e. compiler-inserted or hand-placed inside a phase), with the number of LDS-DMA
issues / transposing LDS reads / global loads / stores / MFMAs seen since the previous line printed.
usage: isa_waits.py file.s <mangled-name-prefix>"""
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or l.startswith(prefix) and ":" in l[:len(prefix) + 200])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
cnt = dict(dma=0, tr=0, ld=0, st=0, mfma=0, dsr=0)
def flush():
    s = " ".join(f"{k}={v}" for k, v in cnt.items() if v)
    for k in cnt: cnt[k] = 0
    return s
i = 0
n = len(body)
while i < n:
    l = body[i].strip()
    if "offen lds" in l or "lds" in l.split(";")[0].split()[-1:] and l.startswith("buffer_load"): cnt["dma"] += 1
    elif l.startswith("ds_read_b64_tr"): cnt["tr"] += 1
    elif l.startswith("ds_read"): cnt["dsr"] += 1
    elif l.startswith("buffer_load"): cnt["ld"] += 1
    elif l.startswith("buffer_store"): cnt["st"] += 1
    elif l.startswith("v_mfma"): cnt["mfma"] += 1
    elif l.startswith("s_waitcnt") and "vmcnt" in l:
        nxt = next((body[j].strip() for j in range(i + 1, min(i + 4, n)) if body[j].strip() and not body[j].strip().startswith(";")), "")
        if nxt.startswith("s_barrier"):
            pass                                    # one case of a counted-barrier ladder
        else:
            print(f"{i:6d}  [{flush()}]  {l}")
    elif l.startswith("s_barrier"):
        m = re.search(r"vmcnt\((\d+)\)", body[i - 1]) or re.search(r"vmcnt\((\d+)\)", body[i - 2])
        if m and m.group(1) != "0" and int(m.group(1)) not in (1,) : pass
    elif l.startswith("s_cbranch") or l.startswith(".LBB"):
        pass
    i += 1
print("tail:", flush(), " lines:", n)
