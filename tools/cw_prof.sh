#!/bin/bash
# per-kernel times of one all-tests run (rocprofv3 --kernel-trace --stats): CFG DT SPREAD
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-ragged}; DT=${2:-i16}; S=${3:-200}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/cwprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cwprof -- python3 $R/bench.py --config $CFG ${MODE---all-tests} --positions 2000000 --dtype $DT --spread $S --steps 5 --warmup 2 --no-cpu --no-side --no-host-path > /tmp/cwprof.log 2>&1
F=$(find /tmp/cwprof -name "*kernel_stats.csv" | head -1); [ -z "$F" ] && { tail -20 /tmp/cwprof.log; find /tmp/cwprof | head; exit 1; }
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:14]:
    print('%-110s calls %5s  avg %10.1f us  total %8.2f ms  %5.1f%%' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
PY
