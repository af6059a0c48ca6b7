#!/bin/bash
# usage: bash tools/kres.sh  -- VGPR / SGPR / scratch / LDS / occupancy of every decode kernel (compiler view, no GPU needed)
cd "$(dirname "$0")/../auroralib/compression_amd/csrc"
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../../include -I. -Wno-unused-function -Wno-inline-asm -x hip -c alz_kernels.hip -o /tmp/kres.o --cuda-device-only -Rpass-analysis=kernel-resource-usage ${ALZ_EXTRA_FLAGS:-} 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k in ('TotalSGPRs','VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(r'\s'+k+r': (\d+)',l)
        if m and cur is not None: cur[k.split(' ')[0]]=int(m.group(1))
for r in rows:
    n=r['name']; m=re.search(r'alz_decode_(\w+?)_kernelILi(\d+)',n)
    tag=(m.group(1)+':'+m.group(2)) if m else n[:40]
    if 'ELi8192' in n: tag+=':8k'
    print('%-16s vgpr %3d sgpr %3d scratch %3d lds %6d occ %d' % (tag, r.get('VGPRs',-1), r.get('TotalSGPRs',-1), r.get('ScratchSize',-1), r.get('LDS',-1), r.get('Occupancy',-1)))
"
