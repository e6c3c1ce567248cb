#!/bin/bash
# tools/coverage_sweep.sh: all tests (and KS only) on event-like int16 rows (sigma 0.2) against the coverage n v n, ~9.2e8 samples per group per pass;
# counting forms on and off.  One line per run: positions/s, samples/s, form share.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for MODE in ${MODES:-alltests ecoli}; do
for N in ${COVERAGES:-10 20 40 64 65 100 128 129 200 255 256 257 300 400 512 513 700 1000 1024 1025 1500 2048}; do
  P=$(( 920000000 / N ))
  for OFF in 0 1; do
    NMOD_NO_COUNTING=$OFF python3 $R/bench.py --config $MODE --n0 $N --n1 $N --positions $P --dtype ${DT:-i16} --spread 200 --steps 5 --warmup 2 --no-cpu --no-side --no-host-path --side-file /tmp/sweep_side.json 2>/tmp/sweep_err.txt | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %4d v %4d counting=%s  %.4g pos/s  %.4g samples/s  K1 %.3f ms  verify %s  %s' % ('$MODE', $N, $N, 'off' if $OFF else 'on ', d['value'], d['value'] * 2 * $N, d['roofline']['kernel_avg_ms'], d['verify']['ok'], d['form_share']))" || tail -3 /tmp/sweep_err.txt
  done
done
done
