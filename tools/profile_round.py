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

CONFIGS = [('ks_f32', []), ('all_f32', ['--config', 'alltests']), ('ks_i16', ['--dtype', 'i16']),
           ('all_i16', ['--config', 'alltests', '--dtype', 'i16']), ('ks_f32_realties', ['--ties', 'real']),
           ('ks_f32_rationald', ['--rational-d']),
           ('all_f32_spread200', ['--config', 'alltests', '--spread', '200']), ('all_i16_spread200', ['--config', 'alltests', '--spread', '200', '--dtype', 'i16']),
           ('ks_f32_spread200', ['--spread', '200']), ('ks_i16_spread200', ['--spread', '200', '--dtype', 'i16']),
           # round 6: event-like rows with 10 per mille outliers (mis-segmented reads over +-5 units)
           ('all_i16_spread200_outl1', ['--config', 'alltests', '--spread', '200', '--dtype', 'i16', '--outliers', '1']),
           ('all_i16_spread200_outl10', ['--config', 'alltests', '--spread', '200', '--dtype', 'i16', '--outliers', '10']),
           # round 6: both groups above 1 024 samples (the value-domain counting form, rank_count_value.hpp)
           ('all_i16_2048v2048_spread200', ['--config', 'alltests', '--n0', '2048', '--n1', '2048', '--positions', '450000', '--spread', '200', '--dtype', 'i16']),
           ('ks_i16_2048v2048_spread200', ['--n0', '2048', '--n1', '2048', '--positions', '450000', '--spread', '200', '--dtype', 'i16']),
           ('all_i16_2048v2048_spread200_outl10', ['--config', 'alltests', '--n0', '2048', '--n1', '2048', '--positions', '450000', '--spread', '200', '--dtype', 'i16', '--outliers', '10'])]
if os.environ.get('NMOD_PROFILE_RAGGED'):  # configs[4] (47 GB of samples, minutes per pass): only on request
    CONFIGS += [('ragged_all_f32', ['--config', 'ragged', '--all-tests', '--steps', '3', '--warmup', '1']),
                ('ragged_all_f32_realties', ['--config', 'ragged', '--all-tests', '--ties', 'real', '--steps', '3', '--warmup', '1']),
                ('ragged_all_i16', ['--config', 'ragged', '--all-tests', '--dtype', 'i16', '--steps', '3', '--warmup', '1']),
                ('ragged_ks_f32', ['--config', 'ragged', '--steps', '3', '--warmup', '1']),
                ('chr20_ks_f32', ['--config', 'chr20', '--steps', '3', '--warmup', '1']),
                ('ragged_all_i16_spread200', ['--config', 'ragged', '--all-tests', '--dtype', 'i16', '--spread', '200', '--steps', '3', '--warmup', '1']),
                ('ragged_all_f32_spread200', ['--config', 'ragged', '--all-tests', '--spread', '200', '--steps', '3', '--warmup', '1']),
                ('chr20_all_i16_spread200', ['--config', 'chr20', '--all-tests', '--dtype', 'i16', '--spread', '200', '--steps', '3', '--warmup', '1']),
                ('chr20_all_f32_spread200', ['--config', 'chr20', '--all-tests', '--spread', '200', '--steps', '3', '--warmup', '1']),
                ('ragged_ks_i16_spread200', ['--config', 'ragged', '--dtype', 'i16', '--spread', '200', '--steps', '3', '--warmup', '1']),
                ('chr20_ks_i16_spread200', ['--config', 'chr20', '--dtype', 'i16', '--spread', '200', '--steps', '3', '--warmup', '1']),
                ('chr20_ks_f32_spread200', ['--config', 'chr20', '--spread', '200', '--steps', '3', '--warmup', '1']),
                ('ragged_all_i16_spread200_outl1', ['--config', 'ragged', '--all-tests', '--dtype', 'i16', '--spread', '200', '--outliers', '1', '--steps', '3', '--warmup', '1']),
                ('ragged_all_i16_spread200_outl10', ['--config', 'ragged', '--all-tests', '--dtype', 'i16', '--spread', '200', '--outliers', '10', '--steps', '3', '--warmup', '1']),
                ('ragged_all_f32_spread200_outl10', ['--config', 'ragged', '--all-tests', '--spread', '200', '--outliers', '10', '--steps', '3', '--warmup', '1']),
                ('chr20_all_i16_spread200_outl10', ['--config', 'chr20', '--all-tests', '--dtype', 'i16', '--spread', '200', '--outliers', '10', '--steps', '3', '--warmup', '1'])]
if len(sys.argv) > 2:                      # python3 tools/profile_round.py r3 ks_f32,all_f32
    CONFIGS = [c for c in CONFIGS if c[0] in sys.argv[2].split(',')]
K1_NAMES = ('ks_rank_kernel', 'rank_hist_kernel', 'rank_pair_kernel', 'big_rank_kernel', 'big_hist_kernel', 'rank_count_kernel', 'rank_count_wide_kernel', 'rank_count_value_kernel')
PMC_GROUPS = [
    ['FETCH_SIZE'], ['WRITE_SIZE'],
    ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY'],
    ['SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_SCA', 'SQ_WAVES', 'GRBM_GUI_ACTIVE'],
]


def run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)


SIDE = '/tmp/prof_side.json'


def bench_line(extra, steps=20, warmup=5, keep_side=None):
    """the LAST stdout line (the compact record); the verbose record + side legs (bench.py --side-file) go to `keep_side`"""
    r = run(['python3', BENCH, '--steps', str(steps), '--warmup', str(warmup), '--side-file', SIDE] + extra)      # (a later --steps in `extra` wins)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if keep_side and os.path.exists(SIDE):
        shutil.copy(SIDE, keep_side)
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
    line = bench_line(extra + (['--no-cpu', '--no-side', '--no-host-path'] if cfg != 'ks_f32' else []),     # the headline: the full default line ...
                      keep_side=os.path.join(OUT, '%s_bench_side_%s.json' % (TAG, cfg)) if cfg == 'ks_f32' else None)   # ... and its side file
    json.dump(line, open(os.path.join(OUT, '%s_bench_%s.json' % (TAG, cfg)), 'w'))
    # kernel trace + stats
    d = '/tmp/prof_%s_trace' % cfg
    shutil.rmtree(d, ignore_errors=True)
    r = run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '--', 'python3', BENCH, '--steps', '10', '--warmup', '3', '--no-cpu', '--no-side', '--no-host-path', '--side-file', SIDE] + extra)
    under = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    for f in glob.glob(d + '/*/*kernel_stats.csv'):
        shutil.copy(f, os.path.join(OUT, '%s_%s_kernel_stats.csv' % (TAG, cfg)))
    if under:
        open(os.path.join(OUT, '%s_%s_bench_under_rocprof.json' % (TAG, cfg)), 'w').write(under[-1] + '\n')
    # PMC passes
    summary = ['rocprofv3 PMC summary, %s, bench.py %s (%s), library sha256[:16] %s' % (TAG, ' '.join(extra) or '(default)', line.get('config', {}).get('workload', '?'), SHA),
               'separate --pmc passes; per-launch means of the K1 kernel(s); FETCH_SIZE / WRITE_SIZE in KiB as reported']
    means = {}
    for gi, group in enumerate(PMC_GROUPS):
        d = '/tmp/prof_%s_pmc%d' % (cfg, gi)
        shutil.rmtree(d, ignore_errors=True)
        run(['rocprofv3', '--pmc'] + group + ['--output-format', 'csv', '-d', d, '--', 'python3', BENCH] + extra + ['--steps', '2', '--warmup', '1', '--no-cpu', '--no-side', '--no-host-path', '--side-file', SIDE])
        vals, meta = k1_rows(d + '/*/*counter_collection.csv', 'Counter_Name')
        # per bench step: 4 passes of the hot path run under the profiler (verify, warm-up, 2 timed); a ragged pass is
        # many size-class launches, so the counters are summed over the K1 kernels and divided by the passes
        passes = 4
        for (kname, cname), v in sorted(vals.items()):
            means[cname] = means.get(cname, 0.0) + sum(v) / passes
            summary.append('  %-46s %-24s launches=%d sum/pass=%.6g' % (kname, cname, len(v), sum(v) / passes))
        if meta and gi == 2:
            summary.append('  (grid, LDS bytes per block, VGPRs, SGPRs) = %r' % (meta,))
    npos = int(line.get('roofline', {}).get('positions_per_launch', 4_600_000))
    key = line.get('roofline', {}).get('profile_key', cfg)
    ent = {'lib_sha16': SHA, 'source': 'profiles/%s_%s_pmc_summary.txt' % (TAG, cfg)}
    if 'FETCH_SIZE' in means and 'WRITE_SIZE' in means:
        hbm = means['FETCH_SIZE'] * 1024 * 2 + means['WRITE_SIZE'] * 1024
        summary.append('HBM traffic per K1 launch (FETCH_SIZE x2: 64 B counted per 128-B request on gfx950 streaming reads, + WRITE_SIZE): %.4g B' % hbm)
        ent['hbm_bytes_per_launch'] = hbm
    if 'SQ_INSTS_VALU' in means:
        ent['valu_instr_per_position'] = means['SQ_INSTS_VALU'] / npos
        summary.append('VALU instructions per position: %.1f' % (means['SQ_INSTS_VALU'] / npos))
    if 'SQ_ACTIVE_INST_VALU' in means and 'GRBM_GUI_ACTIVE' in means:
        ent['valu_issue_util'] = means['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * means['GRBM_GUI_ACTIVE'] / 8)
        summary.append('VALU issue utilisation: SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = %.3f'
                       % ent['valu_issue_util'])
    traffic[key] = ent
    if 'SQ_LDS_BANK_CONFLICT' in means and 'SQ_LDS_IDX_ACTIVE' in means:
        summary.append('LDS bank-conflict share of LDS cycles: %.2f' % (means['SQ_LDS_BANK_CONFLICT'] / means['SQ_LDS_IDX_ACTIVE']))
    open(os.path.join(OUT, '%s_%s_pmc_summary.txt' % (TAG, cfg)), 'w').write('\n'.join(summary) + '\n')
    print('\n'.join(summary[-5:]))
    print(cfg, 'value %.4g  K1 %.3f ms  frac %.3f' % (line.get('value', 0), line.get('roofline', {}).get('kernel_avg_ms', 0), line.get('roofline', {}).get('frac', 0)))
merged = {}
for src in (os.path.join(ROOT, 'profiles', 'traffic.json'), os.path.join(OUT, 'traffic.json')):
    try:
        merged.update({k: v for k, v in json.load(open(src)).items() if v.get('lib_sha16') == SHA})   # same binary only
    except Exception:
        pass
merged.update(traffic)
json.dump(merged, open(os.path.join(OUT, 'traffic.json'), 'w'), indent=1)
