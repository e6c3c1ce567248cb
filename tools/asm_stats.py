#!/usr/bin/env python3
"""Static instruction mix of one kernel in a device assembly file (hipcc --offload-device-only -S):
python3 tools/asm_stats.py file.s <mangled-name-substring>  -> per basic block: VALU / SALU / LDS / VMEM counts."""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(pat) + r'\S*:', l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur = [], ['entry', {}]
def kind(op):
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'
for l in lines[start + 1:end]:
    l = l.split(';')[0].rstrip()
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), {}]; continue
    t = l.strip().split()
    if not t or t[0].startswith('.'): continue
    k = kind(t[0]); cur[1][k] = cur[1].get(k, 0) + 1
    cur[1].setdefault('ops', {}); cur[1]['ops'][t[0]] = cur[1]['ops'].get(t[0], 0) + 1
blocks.append(cur)
tot = {}
for name, c in blocks:
    if sum(v for k, v in c.items() if k != 'ops') >= int(sys.argv[3]) if len(sys.argv) > 3 else 20:
        top = sorted(c.get('ops', {}).items(), key=lambda kv: -kv[1])[:8]
        print('%-12s valu %4d salu %4d lds %3d vmem %2d  %s' % (name, c.get('valu', 0), c.get('salu', 0), c.get('lds', 0), c.get('vmem', 0), ' '.join('%s:%d' % kv for kv in top)))
    for k, v in c.items():
        if k != 'ops': tot[k] = tot.get(k, 0) + v
print('total', tot)
