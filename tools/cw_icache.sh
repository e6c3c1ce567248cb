#!/bin/bash
# tools/cw_icache.sh: instruction-cache and wait counters of rank_count_wide_kernel (separate --pmc passes) on the ragged preset and the
# 500 v 500 shape, int16 event-like rows, per pass of 2 M positions (GPU box; prints to stdout)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for ARGS in "--config ragged --positions 2000000 --dtype i16 --spread 200" "--config chr20 --positions 2000000 --dtype i16 --spread 200"; do
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS"; do
  D=/tmp/cwpmc; rm -rf $D
  rocprofv3 --pmc $grp --output-format csv -d $D -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-side --no-host-path > /dev/null 2> $D.err
  python3 - "$D" <<'PY'
import csv,glob,collections,sys
d=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rank_count_wide' in r['Kernel_Name']:
            d[r['Counter_Name']]+=float(r['Counter_Value'])
for k in sorted(d): print('  %-24s %.4g per pass' % (k, d[k]/4.0))
if not d: print(open(sys.argv[1]+'.err').read()[-400:])
PY
done
done
