#!/bin/bash
# tools/outlier_sweep.sh: all tests on event-like int16 rows (sigma 0.2) against the share of outliers (reads anywhere in +-5 units), configs[2] rows
# (200 v 200), the ragged preset (configs[4], ~1 131 v ~57) and 500 v 500, counting forms on (default) and off.  One line per run.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for CFG in ${CONFIGS:-alltests ragged chr20}; do
  EXTRA="--all-tests --positions ${POS:-2000000}"; [ $CFG = alltests ] && EXTRA=""
  for O in ${OUTLIERS:-0 1 10 30 50 100}; do
    for OFF in 0 1; do
      NMOD_NO_COUNTING=$OFF python3 $R/bench.py --config $CFG $EXTRA --dtype i16 --spread 200 --outliers $O --steps 5 --warmup 2 --no-cpu --no-side --no-host-path --side-file /tmp/sweep_side.json 2>/tmp/sweep_err.txt | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s outliers %3d permille counting=%s  %.4g pos/s  K1 %.3f ms  verify %s  %s' % ('$CFG', $O, 'off' if $OFF else 'on ', d['value'], d['roofline']['kernel_avg_ms'], d['verify']['ok'], d['form_share']))" || tail -5 /tmp/sweep_err.txt
    done
  done
done
