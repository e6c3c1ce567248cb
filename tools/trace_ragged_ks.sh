#!/bin/bash
# per-launch trace of one ragged all-tests step (bench.py --config ragged): tools/trace_ragged.sh [bench args]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p5
rocprofv3 --kernel-trace --output-format csv -d /tmp/p5 -- python3 $R/bench.py --config ragged --steps 2 --warmup 1 --no-cpu "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,re
rows=[]
for f in glob.glob('/tmp/p5/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append(r)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'classify' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
tot=0
for r in rows[idx:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    name=re.sub(r'^void nmod::|\(nmod::RankStatsArgs\)|\(nmod::\w+\)','',r['Kernel_Name'])[:52]
    if d>=0.05: print('%8.3f ms +%7.3f  grid %-8s lds %-6s %s'%((int(r['Start_Timestamp'])-t0)/1e6,d,r.get('Grid_Size_X',r.get('Grid_Size','')),r.get('LDS_Block_Size',''),name))
    tot+=d
print('sum of kernel durations %.3f ms, span %.3f ms'%(tot,(int(rows[-1]['End_Timestamp'])-t0)/1e6))
PY
