#!/bin/bash
# one PMC pass (instruction counts) + one plain bench run; prints K1 per-launch means
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_${1:-x}; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu "$@" > /dev/null 2> $OUT/p.err
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/p/*/*counter_collection.csv')):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(t in r['Kernel_Name'] for t in ('rank_', 'finalize', 'combine')):
            d[(r['Kernel_Name'][:48],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(d.items()): print('  %-50s %-22s %.6g'%(k[0],k[1],sum(v)/len(v)))
PY
python3 $R/bench.py --no-cpu "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('M/s %.1f  ms/step %.3f  K1 ms %.3f'%(d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
