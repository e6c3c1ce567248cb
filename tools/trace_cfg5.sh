#!/bin/bash
# tools/trace_cfg5.sh (round 2): every launch of one configs[4] all-tests step with its duration (profiles/r2_cfg5_kernel_trace.txt)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p5
rocprofv3 --kernel-trace --output-format csv -d /tmp/p5 -- python3 $R/tests/bench_cfg.py --all-tests > /dev/null 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob('/tmp/p5/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append(r)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find the last classify_kernel
idx=[i for i,r in enumerate(rows) if 'classify' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:idx+32]:
    print('%8.3f ms +%7.3f  grid %-8s lds %-6s %s'%((int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,r.get('Grid_Size_X',r.get('Grid_Size','')),r.get('LDS_Block_Size',r.get('LDS_Block_Size_v','')),r['Kernel_Name'][:70]))
PY
