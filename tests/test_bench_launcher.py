"""CPU: `python3 bench.py --gpus N` starts its own ranks.  The children of --launch-only print
their rank and exit before any GPU call, so the launcher path runs in a container without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e,
                          timeout=timeout)


def test_gpus2_spawns_two_fresh_ranks():
    r = _run(['--gpus', '2', '--launch-only', '--config', 'chr20'])
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert sorted(x['rank'] for x in recs) == [0, 1]
    assert all(x['world_size'] == 2 and x['launch_only'] for x in recs)
    assert all(x['master'].startswith('127.0.0.1:') for x in recs)
    assert r.stdout.rstrip().splitlines()[-1].startswith('{')            # a JSON line is the last line of stdout


def test_chr20_on_8_ranks_plan_fits_the_gpus():
    """BASELINE configs[3] in its stated form (64 M positions x 500 v 500 over 8 GPUs) has never run on hardware; what can be
    checked without it: 8 fresh ranks start, each plans 8 M positions of its own (+ halos), 32 GB of samples, and the total
    it will hold — rows, workspace, results, the gathered full-length tracks — fits one MI355X's 288 GB."""
    r = _run(['--gpus', '8', '--launch-only', '--config', 'chr20'], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')), key=lambda x: x['rank'])
    assert [x['rank'] for x in recs] == list(range(8)) and all(x['world_size'] == 8 for x in recs)
    plans = [x['plan'] for x in recs]
    assert all(p['positions_total'] == 64_000_000 and p['chunks'] == 4 and p['block'] == 2_000_000 for p in plans)
    assert sum(p['positions_own'] for p in plans) == 64_000_000 and all(p['positions_own'] == 8_000_000 for p in plans)
    assert all(8_000_000 < p['positions_with_halo'] <= 8_000_000 + 4 * 2 * 2 for p in plans)      # +-nb per block
    assert all(31.9e9 < p['sample_bytes'] < 32.1e9 for p in plans)
    assert all(p['fits_288GB'] and p['device_bytes'] < 40e9 for p in plans)


def test_ragged_plan_uses_the_position_keyed_sizes():
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    a = argparse.Namespace(config='ragged', all_tests=False, n0=0, n1=0, positions=200_000, dtype='f32', chunks=0, strong=False)
    plans = [bench.rank_plan(a, 2, r) for r in (0, 1)]
    assert all(p['positions_total'] == 400_000 and p['chunks'] == 4 for p in plans)
    tot = sum(int(bench.ragged_sizes(bench.SEED, 0, 400_000, g).sum()) for g in (0, 1)) * 4
    # the two ranks hold every row once, + the halo rows twice
    assert tot <= sum(p['sample_bytes'] for p in plans) <= tot + 8 * 2 * 2 * (4000 + 400) * 4


def test_ragged_on_8_ranks_plan_is_balanced():
    """`bench.py --gpus 8 --config ragged --launch-only`: eight fresh ranks, each planning its blocks of the 80 M-position ragged
    genome from the position-keyed sizes — the block-cyclic deal keeps the ranks' sample bytes (the work: K1 is per sample) within
    2 % of each other, and every rank fits its GPU"""
    r = _run(['--gpus', '8', '--config', 'ragged', '--launch-only'], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert sorted(x['rank'] for x in recs) == list(range(8))
    plans = [x['plan'] for x in sorted(recs, key=lambda x: x['rank'])]
    assert all(p['positions_total'] == 80_000_000 and p['chunks'] == 4 for p in plans)
    work = [p['sample_bytes'] for p in plans]
    assert max(work) / min(work) <= 1.02, work
    assert all(p['fits_288GB'] for p in plans) and 40e9 < min(work) and max(work) < 50e9


def test_under_an_external_launcher_it_is_a_rank():
    r = _run(['--gpus', '2', '--launch-only'], env={'RANK': '1', 'WORLD_SIZE': '2', 'LOCAL_RANK': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['rank'] == 1 and rec['world_size'] == 2                    # no second level of children
    r = _run(['--gpus', '4', '--launch-only'], env={'RANK': '0', 'WORLD_SIZE': '2', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_a_failed_rank_is_reported_not_retried():
    """without a GPU every rank fails at its first device call: the parent exits non-zero and prints no record"""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this box could run the ranks')
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--positions', '64', '--no-cpu'], timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert 'a rank failed' in r.stderr


def test_presets_name_the_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    assert set(bench.PRESETS) == {'ecoli', 'alltests', 'chr20', 'ragged'}
    assert bench.PRESETS['chr20']['positions'] == 8_000_000 and bench.PRESETS['chr20']['n0'] == 500
    assert bench.PRESETS['ragged']['layout'] == 'csr'
    import numpy as np
    a = bench.ragged_sizes(1, 0, 100_000, 0); b = bench.ragged_sizes(1, 0, 100_000, 1)
    assert a.min() >= 5 and a.max() <= 4000 and b.min() >= 5 and b.max() <= 400
    assert 900 < np.median(a) < 1100 and 45 < np.median(b) < 55
    # a function of the global position: a halo position gets the same size on every rank
    assert np.array_equal(bench.ragged_sizes(1, 5000, 100, 0), a[5000:5100])


def test_cpu_leg_generator_is_the_device_generator():
    """bench.synth_rows (what the reference-shaped CPU leg's children run on) restates the device generator exactly as
    tests/helpers.synth_ref does (which the -m gpu tests hold against nmod_synth_fill bit for bit)"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    import helpers as H
    for group, i16 in ((0, False), (1, False), (1, True)):
        a = bench.synth_rows(bench.SEED, 9990, 30, group, 57, bench.PLANT_PERIOD, bench.PLANT_SHIFT, i16)
        b = H.synth_ref(bench.SEED, 9990, 30, group, 57, bench.PLANT_PERIOD, bench.PLANT_SHIFT, 'i16' if i16 else 'f32')
        assert a.shape == (30, 57) and np.array_equal(a.reshape(-1), b) and a.dtype == b.dtype
    for group, i16, spread in ((0, False, 200), (1, False, 100), (1, True, 400)):
        a = bench.synth_event_rows(bench.SEED, 9990, 30, group, 57, bench.PLANT_PERIOD, 800, spread, i16)
        b = H.synth_events_ref(bench.SEED, 9990, 30, group, 57, bench.PLANT_PERIOD, 800, spread, 'i16' if i16 else 'f32')
        assert a.shape == (30, 57) and np.array_equal(a, b) and a.dtype == b.dtype


def test_last_line_is_a_compact_record_the_driver_can_keep():
    """the driver keeps an 8 KB tail of stdout and parses the LAST line; round 5's 32 KB line was lost.  The
    compact form of a real default run's full record (tests/golden/bench_full_record.json: the side file of a run on the GPU box)
    stays under 8 000 bytes — under 4 000 for the default legs —, is one line of valid JSON, and carries the contract's keys with
    `roofline` and `cpu_baseline`; a record swollen with extra legs sheds the optional maps instead of growing past the cap."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'bench_full_record.json')))
    side, host_path, drop_in = full['side_legs'], full.get('host_path'), full.get('drop_in_mtest2')
    rec, text = bench.compact_record(full, side, host_path, drop_in, 'bench_side.json')
    assert '\n' not in text and len(text) < 4000, len(text)
    back = json.loads(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'valu', 'verify', 'cpu_baseline', 'build_info', 'side_legs_file', 'form_share'):
        assert k in back, k
    assert back['config']['workload'].startswith('BASELINE.json configs[1]') and 'model' not in back['config']
    r = back['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-5
    assert 'traffic' in r and 'traffic_attached_from' in r and r['kernel'].startswith('ks_rank_kernel')
    cb = back['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['unit'] == 'positions/s' and cb['sample']
    assert back['value'] == float('%.6g' % full['value']) and back['verify']['ok'] is True
    assert set(back['side']) >= {'all_tests', 'int16', 'outliers.all_tests_i16_10_permille', 'ragged_event_outliers.ragged_i16_10_permille'}
    # a run with many more legs: the optional maps go, the contract's keys stay, the line stays under the cap
    fat = dict(side)
    for i in range(400):
        fat['extra_leg_%d' % i] = {'value': 1.0e9 + i, 'verify': {'ok': True}}
    rec2, text2 = bench.compact_record(full, fat, host_path, drop_in, 'bench_side.json')
    assert len(text2) < bench.LAST_LINE_CAP and 'side' not in rec2 and 'roofline' in rec2 and 'cpu_baseline' in rec2
    assert bench.short_build_info(full['build_info']).endswith('experiment switches: none')
