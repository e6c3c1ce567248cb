"""-m gpu: bench.py end to end at small sizes — flag combinations whose legs interact (the rational-D flag through the side legs
and the host-resident leg, the ragged preset with all tests on 3-decimal input, the forced collective with the boundary check):
the line parses, every verification in it holds, the exit code is 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args + ['--steps', '2', '--warmup', '1', '--no-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_rational_d_flag_through_side_and_host_legs():
    d = _bench(['--positions', '150000', '--rational-d'])
    assert d['verify']['ok'] and 'NMOD_FLAG_KS_RATIONAL_D' in d['config']['ks_d']
    assert all(d[k]['verify']['ok'] for k in ('all_tests', 'int16', 'real_ties'))
    hp = d['host_path']
    assert all(v.get('equals_device_resident_pass', v.get('equals_int16_pass')) for v in hp.values() if isinstance(v, dict))
    assert 'float64_grid' in hp
    assert d['roofline']['bound'] == 'valu-issue' and d['build_info'].startswith('arch=gfx950')


def test_default_line_carries_every_leg():
    d = _bench(['--positions', '120000', '--side-legs', 'all_tests,int16,rational_d,real_ties,real_spread'])
    assert d['verify']['ok'] and 'library default' in d['config']['ks_d']
    for k in ('all_tests', 'int16', 'rational_d', 'real_ties'):
        assert d[k]['verify']['ok'] and d[k]['value'] > 0 and 0 < d[k]['roofline_frac'] < 1, k
    rs = d['real_spread']
    for k in ('all_tests_f32_sigma_0.2', 'all_tests_i16_sigma_0.2', 'ks_f32_sigma_0.2', 'ks_i16_sigma_0.2', 'all_tests_f32_sigma_0.1',
              'all_tests_i16_sigma_0.1', 'all_tests_f32_sigma_0.4', 'all_tests_i16_sigma_0.4'):
        assert rs[k]['verify']['ok'] and rs[k]['value'] > 0, k
    assert d['host_path']['pageable_float32']['chunks'] >= 1 and d['host_path']['pinned_h2d_GBps'] > 1


def test_event_like_main_configuration():
    """`--spread 200`: the main workload on event-like rows (both dtypes, all tests): the counting form, verified like any other line"""
    for dt in ('f32', 'i16'):
        d = _bench(['--config', 'alltests', '--spread', '200', '--dtype', dt, '--positions', '200000', '--no-side', '--no-host-path'])
        assert d['verify']['ok'] and d['config']['spread_milli'] == 200 and 'rank_count_kernel' in d['roofline']['kernel']
        assert d['verify']['max_abs_err_ks_d'] == 0.0 and d['verify']['max_abs_err_mwu_u'] == 0.0


def test_event_like_ragged_and_chr20_shapes():
    """`--spread 200` on the ragged preset and the chr20 shape, all tests: the counting form for any coverage, verified against the oracle"""
    for cfg, dt, mode in (('ragged', 'i16', ['--all-tests']), ('ragged', 'f32', ['--all-tests']), ('chr20', 'i16', ['--all-tests']), ('chr20', 'i16', []), ('ragged', 'f32', [])):
        d = _bench(['--config', cfg, '--spread', '200', '--dtype', dt, '--positions', '60000', '--no-side', '--no-host-path'] + mode)
        assert d['verify']['ok'] and d['config']['spread_milli'] == 200 and 'rank_count_wide_kernel' in d['roofline']['kernel'], (cfg, dt, mode)
        assert d['verify']['max_abs_err_ks_d'] == 0.0 and (not mode or d['verify']['max_abs_err_mwu_u'] == 0.0)


def test_ragged_all_tests_grid_input_and_forced_collective():
    d = _bench(['--config', 'ragged', '--positions', '60000', '--all-tests', '--ties', 'real'])
    assert d['verify']['ok'] and d['config']['layout'] == 'csr'
    d = _bench(['--force-collective', '--config', 'ragged', '--positions', '60000', '--all-tests', '--chunks', '3'])
    v = d['verify']
    assert v['ok'] and v['gathered_track_equals_local'] and v['block_boundaries_checked'] == 2 and v['block_boundary_positions_differing'] == 0


def test_drop_in_leg_on_the_reference_dict_shape():
    d = _bench(['--side-legs', 'drop_in'])
    di = d['drop_in_mtest2']
    assert d['verify']['ok'] and di['positions'] == 460000
    for shape in ('arrays', 'lists'):
        assert di[shape]['verify_ok'] and di[shape]['table_lines'] == 460000 and di[shape]['mtest2_s'] < 5.0
    assert di['arrays']['first_ranked'] == di['lists']['first_ranked']
    big = di['at_200v200']
    assert big['reads_per_group'] == 200 and big['arrays']['verify_ok'] and big['arrays']['table_lines'] == 460000
    st = big['arrays']['stages']
    assert st['h2d_bytes'] >= 460000 * 400 * 8 and all(st[k] > 0 for k in ('build_csr_s', 'detect_host_s', 'rank_order_s', 'write_table_s'))
