"""VGPRs / LDS / scratch of the kernels in the built library: python tools/kernel_regs.py [name-fragment ...] (mangled-name fragments)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ccnet_amd", "csrc", "libccnet_cca.so")
with tempfile.TemporaryDirectory() as d:
    fat, co = f"{d}/f", f"{d}/c"
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
ks, cur = [], None
for line in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "agpr_count":
        cur = {"agpr": int(v)}
        ks.append(cur)
    elif cur is not None and k in ("name", "vgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_count"):
        cur[k] = v if k == "name" else int(v)
for k in ks:
    n = k.get("name", "")
    if not n.startswith("_ZN3cca") or (sys.argv[1:] and not all(f in n for f in sys.argv[1:])):
        continue
    print(f"vgpr {k.get('vgpr_count'):4d} agpr {k['agpr']:3d} lds {k.get('group_segment_fixed_size'):7d} scratch {k.get('private_segment_fixed_size'):4d} "
          f"spill {k.get('vgpr_spill_count', 0):3d}  {n}")
