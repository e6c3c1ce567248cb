#!/bin/bash
# tools/cw_pmc_ab.sh: counters of rank_count_wide_kernel for every library in LIBS on one box (ragged preset, all tests, int16 event-like
# rows, OUTLIERS per mille): VALU / SALU / LDS instructions, wave cycles, instruction-cache misses — per pass of 2 M positions
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
LIBS=${LIBS:-"$R/nanomod_amd/libnanomod_hip.so $(ls $R/nanomod_amd/exp/*.so 2>/dev/null)"}
for O in ${OUTLIERS:-0}; do
for LIB in $LIBS; do
  export NMOD_HIP_LIB=$LIB
  echo "== $(basename $LIB) outliers $O"
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    D=/tmp/cwpmc; rm -rf $D
    rocprofv3 --pmc $grp --output-format csv -d $D -- python3 $R/bench.py --config ragged --all-tests --positions 2000000 --dtype i16 --spread 200 --outliers $O --steps 2 --warmup 1 --no-cpu --no-side --no-host-path --side-file /tmp/pmc_side.json > /dev/null 2> $D.err
    python3 - "$D" <<'PY'
import csv,glob,collections,sys
d=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rank_count_wide' in r['Kernel_Name']:
            d[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(d): print('  %-24s %.5g per launch (%d launches)' % (k, d[k]/max(n[k],1), n[k]))
if not d: print(open(sys.argv[1]+'.err').read()[-400:])
PY
  done
done
done
