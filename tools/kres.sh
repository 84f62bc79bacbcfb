#!/bin/bash
# Register / LDS / occupancy report of every kernel in rt_core.hip (compiler remarks; no GPU needed).
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 $@ -c raytracing_amd/csrc/rt_core.hip -o /tmp/rt_core_kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for key in ('VGPRs','AGPRs','SGPRs','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','ScratchSize \[bytes/lane\]'):
        m=re.search(key+r': (\d+)',l)
        if m and cur: rows[cur][key.split(' ')[0]]=int(m.group(1))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    name=re.sub(r'\(.*','',name).replace('void rt::','').replace('rt::','')
    print('%-46s VGPR %3d SGPR %3d LDS %6d scratch %3d occ %d'%(name[:46],v.get('VGPRs',-1),v.get('SGPRs',-1),v.get('LDS',-1),v.get('ScratchSize',-1),v.get('Occupancy',-1)))
"
