#!/bin/bash
# tools/cnt_tail_ab.sh: the counting form of the 256-capacity class with its outlier path, on one box: configs[2] (4.6 M x 200 v 200, all
# tests) on event-like rows with 0 / 1 / 10 per mille outliers, int16 and float32, for every library in LIBS (default: the build +
# nanomod_amd/exp/*.so).  One line per run.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
LIBS=${LIBS:-"$R/nanomod_amd/libnanomod_hip.so $(ls $R/nanomod_amd/exp/*.so 2>/dev/null)"}
for DT in ${DTYPES:-i16 f32}; do
  for O in ${OUTLIERS:-0 1 10}; do
    for LIB in $LIBS; do
      NMOD_HIP_LIB=$LIB python3 $R/bench.py --config alltests --dtype $DT --spread 200 --outliers $O --steps 10 --warmup 3 --no-cpu --no-side --no-host-path --side-file /tmp/ab_side.json 2>/tmp/ab_err.txt | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('200v200 %s outliers %2d %-24s %.4g pos/s  K1 %.3f ms  verify %s  %s' % ('$DT', $O, '$(basename $LIB)', d['value'], d['roofline']['kernel_avg_ms'], d['verify']['ok'], d['form_share']))" || tail -5 /tmp/ab_err.txt
    done
  done
done
