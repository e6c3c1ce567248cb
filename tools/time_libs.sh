#!/bin/bash
# tools/time_libs.sh "<bench args>" lib1.so lib2.so ...: timed run only (phase-skip variants give wrong numbers: no verify gate)
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; shift
for lib in "$@"; do
  NMOD_HIP_LIB=$R/$lib python3 $R/bench.py $ARGS --steps 10 --warmup 3 --no-cpu --no-side --no-host-path 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4g pos/s  K1 %.3f ms  verify %s'%('$lib'.split('/')[-1], d['value'], d['roofline']['kernel_avg_ms'], d['verify']['ok']))"
done
