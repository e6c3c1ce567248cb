#!/bin/bash
# tools/cw_tail_ab.sh: the counting form for any coverage with its tail list, on one box: the ragged preset (configs[4]) and the chr20
# shape, all tests, int16 / float32 event-like rows with 0 / 1 / 10 per mille outliers, for every library in LIBS (default: the build +
# nanomod_amd/exp/*.so, made by tools/build_variant.sh).  One line per run: positions/s, K1 ms, form share.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
LIBS=${LIBS:-"$R/nanomod_amd/libnanomod_hip.so $(ls $R/nanomod_amd/exp/*.so 2>/dev/null)"}
for CFG in ${CONFIGS:-ragged}; do
  for DT in ${DTYPES:-i16}; do
    for O in ${OUTLIERS:-0 1 10}; do
      for LIB in $LIBS; do
        NMOD_HIP_LIB=$LIB python3 $R/bench.py --config $CFG --all-tests --positions ${POS:-2000000} --dtype $DT --spread 200 --outliers $O --steps 5 --warmup 2 --no-cpu --no-side --no-host-path --side-file /tmp/ab_side.json 2>/tmp/ab_err.txt | \
          python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s %s outliers %2d %-28s %.4g pos/s  K1 %.3f ms  verify %s  %s' % ('$CFG', '$DT', $O, '$(basename $LIB)', d['value'], d['roofline']['kernel_avg_ms'], d['verify']['ok'], d['form_share']))" || tail -5 /tmp/ab_err.txt
      done
    done
  done
done
