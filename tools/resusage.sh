#!/bin/bash
# summarise -Rpass-analysis=kernel-resource-usage output: name vgpr agpr scratch spill lds
cd "$(dirname "$0")/../ccnet_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I. -Rpass-analysis=kernel-resource-usage cca_api.hip -o /tmp/libccnet_resusage.so 2>&1 | python3 -c "
import sys,re
cur=None;rows={}
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ['VGPRs','AGPRs','ScratchSize \[bytes/lane\]','VGPRs Spill','SGPRs Spill','LDS Size \[bytes/block\]','Occupancy \[waves/SIMD\]']:
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: rows[cur][k.split(' [')[0].replace('\\\\','')]=m.group(1)
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()[:70]
    print(f'{name:70s}', ' '.join(f'{a}={b}' for a,b in v.items()))
"
