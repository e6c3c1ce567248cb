#!/bin/bash
# A/B of the counting form for any coverage (rank_count_wide.hpp) on one box: all tests on the ragged preset (configs[4], ~1 131 v ~57)
# and on the chr20 shape (500 v 500), event-like rows (--spread S) and the unit-variance rows (S = 0), float32 and int16, with the
# form (default) and without (NMOD_NO_COUNT_WIDE=1: bench.py turns the variable into NMOD_FLAG_NO_COUNT_WIDE; the library reads no environment)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for CFG in ${CONFIGS:-ragged chr20}; do
  POS=${POS:-2000000}
  for S in ${SPREADS:-200 0}; do
    for DT in ${DTYPES:-f32 i16}; do
      for OFF in 0 1; do
        NMOD_NO_COUNT_WIDE=$OFF python3 $R/bench.py --config $CFG --all-tests --positions $POS --dtype $DT --spread $S --steps 5 --warmup 2 --no-cpu --no-side --no-host-path 2>/tmp/ab_err.txt | \
          python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s spread %4d %s count_wide=%s  %.4g pos/s  %.3f ms/step  K1 %.3f ms  frac %.3f  verify %s' % ('$CFG', $S, '$DT', 'off' if $OFF else 'on ', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['verify']['ok']))" || tail -5 /tmp/ab_err.txt
      done
    done
  done
done
