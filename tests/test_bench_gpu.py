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
    """the verbose record of a run (the side file: every leg with its verification) after checking the stdout contract: the LAST
    line is the compact record (< 8 000 bytes, the driver keeps an 8 KB tail), every side leg was a line of its own before it"""
    import tempfile
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'NMOD_NO_COUNTING', 'NMOD_NO_COUNT_WIDE'):
        env.pop(k, None)
    with tempfile.TemporaryDirectory() as tmp:
        side = os.path.join(tmp, 'side.json')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args + ['--steps', '2', '--warmup', '1', '--no-cpu', '--side-file', side],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        last = lines[-1]
        assert len(last) < 8000, len(last)
        c = json.loads(last)
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype', 'config', 'roofline', 'verify', 'build_info', 'side_legs_file', 'form_share'):
            assert k in c, k
        assert c['roofline']['bound'] == 'hbm' and 0 < c['roofline']['frac'] < 1 and c['roofline']['kernel_avg_ms'] > 0
        full = json.load(open(side))
    legs = [json.loads(ln)['side_leg'] for ln in lines[:-1] if ln.startswith('{"side_leg"')]
    assert set(full['side_legs']) <= set(legs), (legs, list(full['side_legs']))
    assert full['value'] == pytest.approx(c['value'], rel=1e-5) and full['verify']['ok'] == c['verify']['ok']
    d = dict(full)
    d.update(full['side_legs'])          # (the legs under their names, as the tests below address them)
    d['compact'] = c
    return d


def test_rational_d_flag_through_side_and_host_legs():
    d = _bench(['--positions', '150000', '--rational-d', '--side-legs', 'all_tests,int16,real_ties,host_path'])
    assert d['verify']['ok'] and 'NMOD_FLAG_KS_RATIONAL_D' in d['config']['ks_d']
    assert all(d[k]['verify']['ok'] for k in ('all_tests', 'int16', 'real_ties'))
    hp = d['host_path']
    assert all(v.get('equals_device_resident_pass', v.get('equals_int16_pass')) for v in hp.values() if isinstance(v, dict))
    assert 'float64_grid' in hp
    assert d['roofline']['bound'] == 'valu-issue' and d['build_info'].startswith('arch=gfx950') and d['compact']['roofline']['limited_by'] == 'valu-issue'


def test_default_line_carries_every_leg():
    d = _bench(['--positions', '120000', '--side-legs', 'all_tests,int16,rational_d,real_ties,real_spread,real_spread_sweep,outliers,host_path'])
    assert d['verify']['ok'] and 'library default' in d['config']['ks_d']
    for k in ('all_tests', 'int16', 'rational_d', 'real_ties'):
        assert d[k]['verify']['ok'] and d[k]['value'] > 0 and 0 < d[k]['roofline_frac'] < 1, k
    rs = d['real_spread']
    for k in ('all_tests_f32_sigma_0.2', 'all_tests_i16_sigma_0.2', 'ks_f32_sigma_0.2', 'ks_i16_sigma_0.2', 'all_tests_f32_sigma_0.1',
              'all_tests_i16_sigma_0.1', 'all_tests_f32_sigma_0.4', 'all_tests_i16_sigma_0.4'):
        assert rs[k]['verify']['ok'] and rs[k]['value'] > 0, k
    assert d['host_path']['pageable_float32']['chunks'] >= 1 and d['host_path']['pinned_h2d_GBps'] > 1
    # which K1 form took each leg's positions: continuous rows stay on the sorting forms, event-like rows go to the counting form
    assert d['compact']['form_share'] == {'ks_rank': 1.0, 'counting_rejected': 0.0}
    assert d['all_tests']['form_share'] == {'rank_hist': 1.0, 'counting_rejected': 0.0}
    for k in ('all_tests_f32_sigma_0.2', 'all_tests_i16_sigma_0.2', 'all_tests_i16_sigma_0.1'):
        assert rs[k]['form_share']['rank_count'] >= 0.99, (k, rs[k]['form_share'])
    ol = d['outliers']
    for k in ('all_tests_f32_1_permille', 'all_tests_i16_1_permille', 'all_tests_f32_10_permille', 'all_tests_i16_10_permille'):
        fs = ol[k]['form_share']
        assert ol[k]['verify']['ok'] and ol[k]['value'] > 0 and abs(fs.get('rank_count', 0.0) + fs.get('rank_hist', 0.0) - 1.0) < 1e-6, (k, fs)
    assert d['compact']['side']['outliers.all_tests_i16_10_permille'] == pytest.approx(ol['all_tests_i16_10_permille']['value'], rel=1e-5)


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
    assert d['compact']['drop_in_mtest2']['at_200v200'] > 0
    di = d['drop_in_mtest2']
    assert d['verify']['ok'] and di['positions'] == 460000
    for shape in ('arrays', 'lists'):
        assert di[shape]['verify_ok'] and di[shape]['table_lines'] == 460000 and di[shape]['mtest2_s'] < 5.0
    assert di['arrays']['first_ranked'] == di['lists']['first_ranked']
    big = di['at_200v200']
    assert big['reads_per_group'] == 200 and big['arrays']['verify_ok'] and big['arrays']['table_lines'] == 460000
    st = big['arrays']['stages']
    assert st['h2d_bytes'] >= 460000 * 400 * 2 and all(st[k] > 0 for k in ('build_csr_s', 'detect_host_s', 'rank_order_s', 'write_table_s'))      # (float64 rows on the 0.001 grid: narrowed to int16 on the way)
