#!/bin/bash
# registers / scratch of the K1 instances: tools/kres.sh <dtype 0|1> <all 0|1> [name filter] [extra -D flags]
cd "$(dirname "$0")/../nanomod_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNMOD_INST_DTYPE=$1 -DNMOD_INST_ALL=$2 $4 -Rpass-analysis=kernel-resource-usage -c rank_stats_inst.hip -o /tmp/kres_$1_$2.o 2>&1 \
 | python3 -c "
import re,sys
name=None; rec={}
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)', ln)
    if m: name=re.sub(r'^_ZN4nmod\d+|EvNS_13RankStatsArgsE$','',m.group(1)); rec[name]={}
    for key in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','LDS Size \[bytes/block\]','Occupancy \[waves/SIMD\]'):
        m=re.search(key+r': (\d+)', ln)
        if m and name: rec[name][key.split(' ')[0]]=int(m.group(1))
flt=sys.argv[1] if len(sys.argv)>1 else '.'
for n,r in rec.items():
    if re.search(flt,n): print('%-44s vgpr %3d scratch %4d occ %d'%(n,r.get('VGPRs',-1),r.get('ScratchSize',-1),r.get('Occupancy',-1)))
" "${3:-.}"
