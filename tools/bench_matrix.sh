#!/bin/bash
# smoke matrix of bench.py's flag combinations at small sizes (GPU box): every line must end with rc=0
for a in "--positions 200000" "--positions 200000 --rational-d" "--positions 200000 --dtype i16 --ties real" "--config ragged --positions 100000 --all-tests --ties real" \
         "--config alltests --positions 200000 --dtype i16" "--config chr20 --positions 100000" "--config chr20 --positions 100000 --all-tests" \
         "--config ragged --positions 100000" "--config ragged --positions 100000 --all-tests --dtype i16" "--config ragged --positions 100000 --ties real" "--config ragged --positions 100000 --all-tests --ties real" \
         "--force-collective --positions 200000 --chunks 3" "--force-collective --config ragged --positions 100000 --all-tests" \
         "--positions 200000 --n0 50 --n1 1000 --all-tests" "--positions 100000 --n0 700 --n1 3000" "--strong --positions 300000"; do
  python bench.py $a --steps 2 --warmup 1 --no-cpu > /tmp/bm.json 2> /tmp/bm.err; rc=$?
  python - "$a" $rc <<'PY'
import json,sys
try:
    l=json.loads(open('/tmp/bm.json').read().strip().splitlines()[-1]); print('%-70s rc=%s value %.3g verify %s'%(sys.argv[1], sys.argv[2], l['value'], l['verify']['ok']))
except Exception as e:
    print('%-70s rc=%s NO LINE'%(sys.argv[1], sys.argv[2]), open('/tmp/bm.err').read()[-400:])
PY
done
