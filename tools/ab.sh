#!/bin/bash
# A/B of library builds on one box: tools/ab.sh "<bench args>" lib1.so lib2.so ...   (PMC pass: VALU instructions and
# busy cycles of the K1 kernels per pass, then a plain timed run)
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; shift
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  export NMOD_HIP_LIB=$R/$lib
  D=/tmp/ab_$(basename $lib .so); rm -rf $D
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $D -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-side --no-host-path > /dev/null 2> $D.err
  python3 - "$D" "$lib" <<'PY'
import csv,glob,collections,sys
d=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if any(t in r['Kernel_Name'] for t in ('ks_rank','rank_hist','rank_pair','big_','rank_count')):
            d[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
p=4.0   # passes of the hot path under the profiler: verify + 1 warm-up + 2 timed
if d: print('%-28s VALU/pass %.4g  busy-cycles/pass %.4g  issue-util %.3f  LDS insts %.4g  LDS-active/busy %.3f  bank-conflict share %.2f' % (sys.argv[2], d['SQ_INSTS_VALU']/p, d['GRBM_GUI_ACTIVE']/p, d['SQ_ACTIVE_INST_VALU']*4/(1024*d['GRBM_GUI_ACTIVE']/8), d['SQ_INSTS_LDS']/p, d['SQ_LDS_IDX_ACTIVE']/(256*d['GRBM_GUI_ACTIVE']/8), d['SQ_LDS_BANK_CONFLICT']/max(d['SQ_LDS_IDX_ACTIVE'],1)))
else: print(sys.argv[2], 'no counters', open(sys.argv[1]+'.err').read()[-300:])
PY
  python3 $R/bench.py $ARGS --steps 20 --warmup 5 --no-cpu --no-side --no-host-path 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4g pos/s  K1 %.3f ms  verify %s'%('', d['value'], d['roofline']['kernel_avg_ms'], d['verify']['ok']))"
done
