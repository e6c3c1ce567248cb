#!/bin/bash
# A/B of the counting form on one box: all tests at 200 v 200, event-like rows (--spread S) and the unit-variance rows (S = 0),
# float32 and int16, with the counting form (default) and without (NMOD_NO_COUNTING=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for S in ${SPREADS:-200 100 400 0}; do
  for DT in f32 i16; do
    for OFF in 0 1; do
      NMOD_NO_COUNTING=$OFF python3 $R/bench.py --config alltests --dtype $DT --spread $S --steps 10 --warmup 3 --no-cpu --no-side --no-host-path 2>/tmp/ab_err.txt | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spread %4d %s counting=%s  %.4g pos/s  K1 %.3f ms  frac %.3f  verify %s' % ($S, '$DT', 'off' if $OFF else 'on ', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['verify']['ok']))" || tail -5 /tmp/ab_err.txt
    done
  done
done
