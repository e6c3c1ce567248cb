#!/bin/bash
# tools/quick_bench.sh "<bench args>" ... : one short line per configuration (GPU box)
for a in "$@"; do
  python bench.py $a --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    l=json.loads(t[-1])
    print('%-9s %s %-4s value %.4g k1 %.3f ms frac %.3f verify %s real-ties %.4g' % (l['config']['preset'], 'all' if 'MWU' in l['metric'] else 'ks ', l['dtype'][:3], l['value'], l['roofline']['kernel_avg_ms'], l['roofline']['frac'], l['verify']['ok'], l.get('real_ties',{}).get('value',0)))
except Exception as e:
    print('FAILED', sys.argv[1:], e, t[-3:])
" "$a"
done
