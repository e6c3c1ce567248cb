#!/usr/bin/env python3
"""Round profile (run on the GPU box: python3 tools/profile_round.py r2).

For each bench configuration (KS + Stouffer f32 = the headline, all tests + Fisher f32, KS + Stouffer int16):
  * the plain bench line                                   -> gpurun_out/<tag>_bench_<cfg>.json
  * rocprofv3 --kernel-trace --stats of the same command   -> gpurun_out/<tag>_<cfg>_kernel_stats.csv
  * separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ groups), per-launch means of the K1 kernel
                                                           -> gpurun_out/<tag>_<cfg>_pmc_summary.txt
and profiles-ready traffic records keyed like bench.py's lookup (lib sha + HBM bytes per K1 launch)
                                                           -> gpurun_out/traffic.json
Counters are collected in their own runs, never together with a trace (MI355X_MICROARCH.md, rocprofv3 section).
The profiled program is `python3 bench.py ...` itself (no wrapper between rocprofv3 and the program)."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'rX'
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
os.environ['TMPDIR'] = '/tmp'
os.chdir('/tmp')
BENCH = os.path.join(ROOT, 'bench.py')
LIB = os.path.join(ROOT, 'nanomod_amd', 'libnanomod_hip.so')
SHA = hashlib.sha256(open(LIB, 'rb').read()).hexdigest()[:16]

CONFIGS = [('ks_f32', []), ('all_f32', ['--all-tests']), ('ks_i16', ['--dtype', 'i16'])]
K1_NAMES = ('ks_rank_kernel', 'rank_hist_kernel', 'rank_pair_kernel', 'big_rank_kernel', 'big_hist_kernel')
PMC_GROUPS = [
    ['FETCH_SIZE'], ['WRITE_SIZE'],
    ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY'],
    ['SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_SCA', 'SQ_WAVES', 'GRBM_GUI_ACTIVE'],
]


def run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)


def bench_line(extra, steps=20, warmup=5):
    r = run(['python3', BENCH, '--steps', str(steps), '--warmup', str(warmup)] + extra)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return json.loads(lines[-1]) if lines else {'error': r.stderr[-2000:]}


def k1_rows(pattern, col):
    vals = collections.defaultdict(list)
    meta = None
    for f in glob.glob(pattern):
        for row in csv.DictReader(open(f)):
            if any(n in row['Kernel_Name'] for n in K1_NAMES):
                vals[(row['Kernel_Name'].split('(')[0][-60:], row[col])].append(float(row['Counter_Value']))
                meta = (row.get('Grid_Size'), row.get('LDS_Block_Size'), row.get('VGPR_Count'), row.get('SGPR_Count'))
    return vals, meta


traffic = {}
for cfg, extra in CONFIGS:
    line = bench_line(extra + (['--no-cpu'] if cfg != 'ks_f32' else []))
    json.dump(line, open(os.path.join(OUT, '%s_bench_%s.json' % (TAG, cfg)), 'w'))
    # kernel trace + stats
    d = '/tmp/prof_%s_trace' % cfg
    shutil.rmtree(d, ignore_errors=True)
    r = run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--', 'python3', BENCH, '--steps', '10', '--warmup', '3', '--no-cpu'] + extra)
    under = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    for f in glob.glob(d + '/*/*kernel_stats.csv'):
        shutil.copy(f, os.path.join(OUT, '%s_%s_kernel_stats.csv' % (TAG, cfg)))
    if under:
        open(os.path.join(OUT, '%s_%s_bench_under_rocprof.json' % (TAG, cfg)), 'w').write(under[-1] + '\n')
    # PMC passes
    summary = ['rocprofv3 PMC summary, %s, bench.py %s (4.6 M positions, 200 v 200), library sha256[:16] %s' % (TAG, ' '.join(extra) or '(default)', SHA),
               'separate --pmc passes; per-launch means of the K1 kernel(s); FETCH_SIZE / WRITE_SIZE in KiB as reported']
    means = {}
    for gi, group in enumerate(PMC_GROUPS):
        d = '/tmp/prof_%s_pmc%d' % (cfg, gi)
        shutil.rmtree(d, ignore_errors=True)
        run(['rocprofv3', '--pmc'] + group + ['--output-format', 'csv', '-d', d, '--', 'python3', BENCH, '--steps', '2', '--warmup', '1', '--no-cpu'] + extra)
        vals, meta = k1_rows(d + '/*/*counter_collection.csv', 'Counter_Name')
        for (kname, cname), v in sorted(vals.items()):
            means[cname] = sum(v) / len(v)
            summary.append('  %-46s %-24s launches=%d mean=%.6g' % (kname, cname, len(v), means[cname]))
        if meta and gi == 2:
            summary.append('  (grid, LDS bytes per block, VGPRs, SGPRs) = %r' % (meta,))
    npos = 4_600_000
    if 'FETCH_SIZE' in means and 'WRITE_SIZE' in means:
        hbm = means['FETCH_SIZE'] * 1024 * 2 + means['WRITE_SIZE'] * 1024
        summary.append('HBM traffic per K1 launch (FETCH_SIZE x2: 64 B counted per 128-B request on gfx950 streaming reads, + WRITE_SIZE): %.4g B' % hbm)
        n0 = n1 = 200
        key = '%s_%s_%dv%d_%d' % ('all' if '--all-tests' in extra else 'ks', 'i16' if 'i16' in extra else 'f32', n0, n1, npos)
        traffic[key] = {'lib_sha16': SHA, 'hbm_bytes_per_launch': hbm, 'source': 'profiles/%s_%s_pmc_summary.txt' % (TAG, cfg)}
    if 'SQ_INSTS_VALU' in means:
        summary.append('VALU instructions per position: %.1f' % (means['SQ_INSTS_VALU'] / npos))
    if 'SQ_ACTIVE_INST_VALU' in means and 'GRBM_GUI_ACTIVE' in means:
        summary.append('VALU issue utilisation: SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = %.3f'
                       % (means['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * means['GRBM_GUI_ACTIVE'] / 8)))
    if 'SQ_LDS_BANK_CONFLICT' in means and 'SQ_LDS_IDX_ACTIVE' in means:
        summary.append('LDS bank-conflict share of LDS cycles: %.2f' % (means['SQ_LDS_BANK_CONFLICT'] / means['SQ_LDS_IDX_ACTIVE']))
    open(os.path.join(OUT, '%s_%s_pmc_summary.txt' % (TAG, cfg)), 'w').write('\n'.join(summary) + '\n')
    print('\n'.join(summary[-5:]))
    print(cfg, 'value %.4g  K1 %.3f ms  frac %.3f' % (line.get('value', 0), line.get('roofline', {}).get('kernel_avg_ms', 0), line.get('roofline', {}).get('frac', 0)))
json.dump(traffic, open(os.path.join(OUT, 'traffic.json'), 'w'), indent=1)
