#!/bin/bash
# bench.py on event-like rows (--spread S milli-units) for the all-tests and KS-only modes, float32 and int16:
# tools/spread_sweep.sh [lib.so]   -> one line per configuration (positions/s, K1 ms, verify)
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$1" ] && export NMOD_HIP_LIB=$R/$1
cd /tmp; export TMPDIR=/tmp
for S in 0 100 200 400; do
  for DT in f32 i16; do
    for CFG in alltests ecoli; do
      python3 $R/bench.py --config $CFG --dtype $DT --spread $S --steps 10 --warmup 3 --no-cpu --no-side --no-host-path 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spread %4d %s %-8s %.4g pos/s  K1 %.3f ms  frac %.3f  verify %s  %s' % ($S, '$DT', '$CFG', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['verify']['ok'], d['roofline']['kernel']))"
    done
  done
done
