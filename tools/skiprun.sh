#!/bin/bash
# tools/skiprun.sh (round 3): the bench over phase-skip builds of the library (-DNMOD_SKIP=<bits>, libnanomod_hip_s<bits>.so beside the build) —
# where the instructions of the all-tests kernel go (profiles/HISTORY.md B.3).  Timing only: phase-skip builds give wrong numbers.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/nanomod_amd/libnanomod_hip.so /tmp/base.so
for S in 0 1 32 64 96; do
  if [ $S = 0 ]; then cp /tmp/base.so $R/nanomod_amd/libnanomod_hip.so; else cp $R/nanomod_amd/libnanomod_hip_s$S.so $R/nanomod_amd/libnanomod_hip.so; fi
  rm -rf /tmp/p$S
  rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/p$S -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --all-tests --positions 1000000 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[]
for f in glob.glob('/tmp/p$S/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rank_hist' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
print('SKIP=$S instr/pos %.1f'%(sum(v)/len(v)/1000000))
PY
done
cp /tmp/base.so $R/nanomod_amd/libnanomod_hip.so
