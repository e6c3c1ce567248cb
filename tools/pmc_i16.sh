#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_i16; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $R/tools/bench_i16.py > /dev/null 2> $OUT/p.err
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/p/*/*counter_collection.csv')):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ks_rank' in r['Kernel_Name'] or 'rank_all' in r['Kernel_Name']:
            d[(r['Kernel_Name'][:52],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(d.items()): print('  %-54s %-22s %.6g'%(k[0],k[1],sum(v)/len(v)))
PY
