#!/bin/bash
# where the waves of the K1 kernels wait: one --pmc pass (no trace) over `bench.py $1`, counters summed over the K1 kernels per pass
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; shift
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  D=/tmp/cwpmc; rm -rf $D
  rocprofv3 --pmc $grp --output-format csv -d $D -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-side --no-host-path > /dev/null 2> $D.err
  python3 - "$D" <<'PY'
import csv,glob,collections,sys
d=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if any(t in r['Kernel_Name'] for t in ('ks_rank','rank_hist','rank_pair','big_','rank_count')):
            d[r['Counter_Name']]+=float(r['Counter_Value'])
for k in sorted(d): print('  %-24s %.4g per pass' % (k, d[k]/4.0))
if not d: print(open(sys.argv[1]+'.err').read()[-400:])
PY
done
