"""The simulation repeat loops (nanomod_amd.simulate) against golden runs of the reference's own getGenomeEvents +
mfilter_coverage + mtest2 + getTopRank with the random draws given (oracle/gen_golden_sim.py -> tests/golden/
simulate_repeats.npz).  CPU: the event grouping, the tested-position set and the ranking walk; GPU: the whole repeat."""
import os

import numpy as np
import pytest

import helpers as H

Z = np.load(os.path.join(H.GOLDEN, 'simulate_repeats.npz'))
NAMES = [str(n) for n in Z['names']]


def _pools(device):
    from nanomod_amd import simulate
    return {name: simulate.ReadPool(Z[name + '_chrom'], Z[name + '_strand'], Z[name + '_start'], Z[name + '_off'],
                                    Z[name + '_norm_mean'], Z[name + '_base'], device=device) for name in ('case', 'control')}


def _parts(name, pools):
    case, control = [], []
    for j in range(int(Z[name + '_nsel'])):
        item = (pools[str(Z['%s_sel%d_pool' % (name, j)])], Z['%s_sel%d_idx' % (name, j)])
        (case if str(Z['%s_sel%d_label' % (name, j)]) == 'simulate_case' else control).append(item)
    return case, control


def _opts(name):
    o = {'MinCoverage': 5, 'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv', 'window': 2,
         'RegionRankbyST': 0}
    if name == 'fisher_w5':
        o.update(testMethod='fisher', window=5)
    return o


@pytest.mark.parametrize('name', NAMES)
def test_grouping_position_set_and_rank_walk_cpu(name):
    """no GPU: drawn reads -> per-position groups -> tested positions equal the reference's (set, order, coverages), and
    get_top_rank on the reference's own p-values returns the reference's rank"""
    from nanomod_amd import simulate, detect
    pools = _pools('cpu')
    case, control = _parts(name, pools)
    cs_names = sorted(set(cs for pool, _ in case + control for cs in pool.cs_names))
    g0 = simulate.group_events(case, cs_names); g1 = simulate.group_events(control, cs_names)
    key, base, (sig0, off0), (sig1, off1) = simulate.tested_positions(g0, g1, 5)
    cs = (key >> 40).numpy(); pos = (key & ((1 << 40) - 1)).numpy()
    chrom = np.array([cs_names[c][0] for c in cs]); strand = np.array([cs_names[c][1] for c in cs])
    assert list(chrom) == list(Z[name + '_chrom']) and list(strand) == list(Z[name + '_strand'])
    assert np.array_equal(pos, Z[name + '_pos'])
    assert np.array_equal(np.diff(off0.numpy()), Z[name + '_n0']) and np.array_equal(np.diff(off1.numpy()), Z[name + '_n1'])
    rid = detect.run_ids(chrom, strand, pos)
    order = np.lexsort((Z[name + '_ks_p'], Z[name + '_comb_p']))                       # stable, like sorted() on the tuple
    o = _opts(name)
    assert simulate.get_top_rank(chrom, strand, pos, order, rid, o['neighborPvalues'], o['window']) == int(Z[name + '_rank'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
@pytest.mark.parametrize('device', ['cuda:0', 'cpu'])
def test_repeat_through_the_hip_path(name, device):
    """the whole repeat (device grouping, intersection, HIP tests + combine + ranking, rank walk): the reference's rank
    and p-values; pools resident on the GPU (float64 device tensors) or on the host (staged by the library)"""
    from nanomod_amd import simulate
    pools = _pools(device)
    case, control = _parts(name, pools)
    rank, tab = simulate.run_repeat(case, control, _opts(name))
    assert rank == int(Z[name + '_rank'])
    assert np.array_equal(tab['pos'], Z[name + '_pos'])
    H.assert_close_p(tab['comb_p'], Z[name + '_comb_p'], 1e-9, 'comb_p')
    H.assert_close_p(tab['ks_p'], Z[name + '_ks_p'], 1e-9, 'ks_p')
    assert np.array_equal(tab['mwu_u'], Z[name + '_mwu_u'])


@pytest.mark.gpu
def test_random_repeat_loops_find_the_planted_site():
    """seeded draws (the reference is unseeded: only the distribution is comparable): with a strong shift at the known
    site most repeats rank it first, more modified reads rank it no worse on average, and a seed reproduces its ranks"""
    from nanomod_amd import simulate
    pools = _pools('cuda:0')
    o = _opts('x')
    r1 = simulate.simulate_case_size(pools['case'], pools['control'], 100, 0.5, 12, o, seed=3)
    r2 = simulate.simulate_case_size(pools['case'], pools['control'], 100, 0.5, 12, o, seed=3)
    assert r1 == r2 and len(r1) == 12
    assert sum(1 for r in r1 if 1 <= r <= 3) >= 9
    weak = simulate.simulate_case_size(pools['case'], pools['control'], 25, 0.25, 12, o, seed=4)
    assert np.mean([r if r > 0 else 50 for r in weak]) >= np.mean([r if r > 0 else 50 for r in r1])
    d = simulate.down_sampling(pools['case'], pools['control'], 150, 8, o, seed=5, max_draws=200)
    assert len(d) == 8 and sum(1 for r in d if 1 <= r <= 3) >= 6
