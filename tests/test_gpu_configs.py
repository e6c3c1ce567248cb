"""-m gpu: BASELINE.json configs[2..4] at (or near) their full shapes on one GPU.

Each config is checked three ways: against the C oracle on a sample of its positions (same inputs, copied back),
through size-independent properties over ALL positions (group-swap symmetry, permutation invariance inside a group,
KS-only == all-tests, logical shards == unsharded), and through the planted sites.  The oracle is the checker only."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nm():
    import nanomod_amd
    return nanomod_amd


def _oracle_sample(sig0, sig1, n0, n1, lo, cnt, nb, method, tests=7):
    """oracle on positions [lo, lo + cnt) of fixed-stride device arrays (one contiguous run)"""
    import oracle_c
    a = sig0[lo * n0:(lo + cnt) * n0].cpu().numpy()
    b = sig1[lo * n1:(lo + cnt) * n1].cpu().numpy()
    off0 = np.arange(0, (cnt + 1) * n0, n0, dtype=np.int64)
    off1 = np.arange(0, (cnt + 1) * n1, n1, dtype=np.int64)
    return oracle_c.detect_batch(a, off0, b, off1, np.zeros(cnt, np.int32), nb, 2.0, method, tests=tests, threads=0)


def _cmp_sample(res, exp, lo, cnt, nb, names):
    inner = slice(nb, cnt - nb)            # the sample's edge windows see neighbours the oracle run did not
    for k in names:
        g = res[k][lo:lo + cnt].cpu().numpy()
        if k.endswith('_p'):
            H.assert_close_p(g[inner] if k.startswith('comb') else g, exp[k][inner] if k.startswith('comb') else exp[k], 1e-9, k)
        elif k == 'mwu_u':
            assert np.array_equal(g, exp[k]), k
        elif k == 'ks_d':
            H.assert_close_stat(g, exp[k], 0, 0.0, k)
        elif k == 'comb_st':
            H.assert_close_stat(g[inner], exp[k][inner], 1e-9, 1e-12, k)
        else:
            # t = (mean0 - mean1) / se with |mean| ~ 1e-2, se ~ 0.1: one ulp of a sample (1e-16) in a mean moves t by
            # 1e-15 absolute — both sides sum in different orders, so near t = 0 only the absolute error is meaningful
            H.assert_close_stat(g, exp[k], 1e-11, 2e-14, k)


def test_cfg3_all_tests_fisher_full_size(nm):
    """configs[2]: E. coli 4.6 M x 200 v 200, KS + MWU + Welch-t + Fisher: oracle on three 20 000-position samples
    (start, middle with a planted site, end), group-swap symmetry and the planted sites over all positions"""
    import torch
    L = nm._lib
    P, n, nb = 4_600_000, 200, 2
    dev = 'cuda:0'
    det = nm.DeviceDetector(0, nb=nb, weights_dif=2.0, method='fisher', tests=L.TEST_ALL)
    a = torch.empty(P * n, dtype=torch.float32, device=dev); b = torch.empty(P * n, dtype=torch.float32, device=dev)
    det.synth_fill(a, 11, 0, P, 0, n, 10000, 0.8); det.synth_fill(b, 11, 0, P, 1, n, 10000, 0.8)
    rid = torch.zeros(P, dtype=torch.int32, device=dev)
    r1 = {k: v.clone() for k, v in det.run(a, b, rid, stride0=n, stride1=n, npos=P).items()}
    torch.cuda.synchronize()
    names = ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p')
    cnt = 20000
    for lo in (0, 2_300_000 - 7, P - cnt):
        exp = _oracle_sample(a, b, n, n, lo, cnt, nb, 'fisher')
        # a sample that starts / ends inside the genome has neighbours outside it: compare the inner windows only;
        # at the genome's own ends (lo == 0, lo + cnt == P) the padding is the reference's and must match too
        _cmp_sample(r1, exp, lo, cnt, nb, names)
        if lo == 0:
            H.assert_close_p(r1['comb_p'][:nb].cpu().numpy(), exp['comb_p'][:nb], 1e-9, 'comb_p at the left end')
        if lo + cnt == P:
            H.assert_close_p(r1['comb_p'][P - nb:].cpu().numpy(), exp['comb_p'][cnt - nb:], 1e-9, 'comb_p at the right end')
    r2 = det.run(b, a, rid, stride0=n, stride1=n, npos=P)
    torch.cuda.synchronize()
    for k in ('ks_d', 'ks_p', 'mwu_u', 'mwu_p', 'comb_p', 'comb_st'):
        assert torch.equal(r1[k], r2[k]), k
    planted = torch.arange(10000, P, 10000, device=dev)
    assert float(r1['comb_p'][planted].max().item()) < 1e-10
    assert int(r1['status'].max().item()) == 0


def test_cfg4_shape_500v500_fixed_stride(nm):
    """configs[3]'s per-GPU share at full size: 8 M positions x 500 v 500 reads, fixed stride, KS + Stouffer (32 GB of
    samples): oracle samples, swap symmetry, invariance under a permutation of the reads inside every position,
    all-tests mode agrees on D and p"""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 80 * 2 ** 30:
        pytest.skip('needs ~60 GB of device memory')
    import torch
    L = nm._lib
    P, n, nb = 8_000_000, 500, 2          # chr20's per-GPU share of configs[3]: 2 x 16 GB of float32, element offsets beyond 2^32
    dev = 'cuda:0'
    det = nm.DeviceDetector(0, nb=nb, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    a = torch.empty(P * n, dtype=torch.float32, device=dev); b = torch.empty(P * n, dtype=torch.float32, device=dev)
    det.synth_fill(a, 5, 0, P, 0, n, 10000, 0.8); det.synth_fill(b, 5, 0, P, 1, n, 10000, 0.8)
    rid = torch.zeros(P, dtype=torch.int32, device=dev)
    r1 = {k: v.clone() for k, v in det.run(a, b, rid, stride0=n, stride1=n, npos=P).items()}
    torch.cuda.synchronize()
    cnt = 8000
    for lo in (0, 3_999_990, P - cnt):
        exp = _oracle_sample(a, b, n, n, lo, cnt, nb, 'stouffer', tests=1)
        _cmp_sample(r1, exp, lo, cnt, nb, ('ks_d', 'ks_p', 'comb_st', 'comb_p'))
    r2 = det.run(b, a, rid, stride0=n, stride1=n, npos=P)
    torch.cuda.synchronize()
    for k in ('ks_d', 'ks_p', 'comb_p', 'comb_st'):
        assert torch.equal(r1[k], r2[k]), k
    # reverse the reads of every position of group 1: same multisets, same numbers
    a_rev = a.view(P, n).flip(1).contiguous().view(-1)
    r3 = det.run(a_rev, b, rid, stride0=n, stride1=n, npos=P)
    torch.cuda.synchronize()
    for k in ('ks_d', 'ks_p', 'comb_p'):
        assert torch.equal(r1[k], r3[k]), k
    del a_rev, r2, r3
    # all three tests on the first 300 000 positions: D to one rounding, the KS p-value and the combination to 1e-9
    Q = 300_000
    det_all = nm.DeviceDetector(0, nb=nb, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL)
    ra = det_all.run(a[:Q * n], b[:Q * n], rid[:Q], stride0=n, stride1=n, npos=Q)
    torch.cuda.synchronize()
    assert float((ra['ks_d'] - r1['ks_d'][:Q]).abs().max().item()) <= 0.0
    inner = slice(0, Q - nb)
    assert float(((ra['comb_p'][inner] - r1['comb_p'][:Q][inner]).abs() / r1['comb_p'][:Q][inner]).max().item()) <= 1e-9
    exp = _oracle_sample(a, b, n, n, 0, 4000, nb, 'stouffer')
    _cmp_sample(ra, exp, 0, 4000, nb, ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p'))
    planted = torch.arange(10000, P, 10000, device=dev)
    assert float(r1['comb_p'][planted].max().item()) < 1e-20


@pytest.mark.parametrize('tests_all', [False, True])
def test_cfg5_ragged_lognormal_one_million(nm, tests_all):
    """configs[4]: skewed ragged coverage, SURVEY.md §8(d) sizes (n0 ~ LogNormal(ln 1000, 0.5) in [5, 4000],
    n1 ~ LogNormal(ln 50, 0.5) in [5, 400]), 1 M positions, CSR: oracle on position samples (all size classes occur,
    large positions included), and the size-balanced logical shards of sharding.balanced_bounds reassemble to the
    unsharded tracks bit for bit"""
    import torch
    import oracle_c
    from nanomod_amd import sharding
    L = nm._lib
    P, nb = 1_000_000, 2
    dev = 'cuda:0'
    rng = np.random.default_rng(5)
    n0 = np.clip(np.round(rng.lognormal(np.log(1000), 0.5, P)), 5, 4000).astype(np.int64)
    n1 = np.clip(np.round(rng.lognormal(np.log(50), 0.5, P)), 5, 400).astype(np.int64)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum(n1)
    g = torch.Generator(device=dev); g.manual_seed(9)
    s0 = torch.randn(int(off0[-1]), dtype=torch.float32, device=dev, generator=g)
    s1 = torch.randn(int(off1[-1]), dtype=torch.float32, device=dev, generator=g) + 0.1
    # 3-dp rounding on a slice of the positions: ties inside and across the groups
    cut0, cut1 = int(off0[200_000]), int(off1[200_000])
    s0[:cut0] = torch.round(s0[:cut0] * 100) / 100; s1[:cut1] = torch.round(s1[:cut1] * 100) / 100
    o0 = torch.from_numpy(off0).to(dev); o1 = torch.from_numpy(off1).to(dev)
    rid = torch.as_tensor((np.arange(P) // 50_021).astype(np.int32), device=dev)
    det = nm.DeviceDetector(0, nb=nb, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL if tests_all else L.TEST_KS)
    full = {k: v.clone() for k, v in det.run(s0, s1, rid, off0=o0, off1=o1, max_n0=4000, max_n1=400).items()}
    torch.cuda.synchronize()
    assert int(full['status'].max().item()) == 0
    names = ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p') if tests_all else ('ks_d', 'ks_p')
    for lo, cnt in ((0, 1500), (199_000, 2000), (P - 1500, 1500)):
        a = s0[int(off0[lo]):int(off0[lo + cnt])].cpu().numpy(); b = s1[int(off1[lo]):int(off1[lo + cnt])].cpu().numpy()
        exp = oracle_c.detect_batch(a, off0[lo:lo + cnt + 1] - off0[lo], b, off1[lo:lo + cnt + 1] - off1[lo],
                                    rid[lo:lo + cnt].cpu().numpy(), nb, 2.0, 'stouffer', tests=7 if tests_all else 1, threads=0)
        _cmp_sample(full, exp, lo, cnt, nb, names)
    # the largest positions explicitly (beyond the wave-resident kernels in all-tests mode)
    big = np.argsort(n0)[-40:]
    for i in big[::4]:
        a = s0[int(off0[i]):int(off0[i + 1])].cpu().numpy(); b = s1[int(off1[i]):int(off1[i + 1])].cpu().numpy()
        e = oracle_c.detect_batch(a, np.array([0, len(a)]), b, np.array([0, len(b)]), np.zeros(1, np.int32), 0, 2.0, 'ks',
                                  tests=7 if tests_all else 1, threads=1)
        assert abs(float(full['ks_d'][i].item()) - e['ks_d'][0]) <= 0.0
        assert abs(float(full['ks_p'][i].item()) - e['ks_p'][0]) <= 1e-9 * e['ks_p'][0]
        if tests_all:
            assert float(full['mwu_u'][i].item()) == e['mwu_u'][0]
    # size-balanced logical shards (equal sample counts, not equal position counts) + halo, run one after the other
    G = 4
    got = {k: torch.empty_like(full[k]) for k in ('ks_p', 'comb_p', 'comb_st')}
    work = []
    for r in range(G):
        lo, hi = sharding.balanced_bounds(off0, off1, G, r)
        lo_h, hi_h = sharding.halo_bounds(lo, hi, nb, P)
        work.append(int(off0[hi] - off0[lo] + off1[hi] - off1[lo]))
        part = det.run(s0, s1, rid[lo_h:hi_h], off0=o0[lo_h:hi_h + 1], off1=o1[lo_h:hi_h + 1], max_n0=4000, max_n1=400)
        for k in got:
            got[k][lo:hi] = part[k][lo - lo_h:lo - lo_h + (hi - lo)]
    torch.cuda.synchronize()
    assert max(work) - min(work) <= 2 * 4400
    for k in got:
        assert torch.equal(got[k], full[k]), k


@pytest.mark.parametrize('tests_all', [False, True])
def test_cfg5_full_size_ten_million(nm, tests_all):
    """configs[4] at BASELINE.json's full size on one GPU: 10 M ragged positions, 1.2e10 samples (47 GB of float32 in
    HBM, offsets beyond 2^32 bytes).  Oracle on position slices from the start, the middle and the end, on the largest
    positions, and the swap symmetry of D over the whole batch."""
    import torch
    import oracle_c
    L = nm._lib
    P, nb = 10_000_000, 2
    dev = 'cuda:0'
    if torch.cuda.get_device_properties(0).total_memory < 80 * 2 ** 30:
        pytest.skip('needs ~60 GB of device memory')
    rng = np.random.default_rng(55)
    n0 = np.clip(np.round(rng.lognormal(np.log(1000), 0.5, P)), 5, 4000).astype(np.int64)
    n1 = np.clip(np.round(rng.lognormal(np.log(50), 0.5, P)), 5, 400).astype(np.int64)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum(n1)
    assert off0[-1] * 4 > 2 ** 32
    g = torch.Generator(device=dev); g.manual_seed(10)
    s0 = torch.empty(int(off0[-1]), dtype=torch.float32, device=dev)
    step = 1 << 30
    for lo in range(0, s0.numel(), step):                       # (piecewise: the generator's own temporaries stay small)
        s0[lo:lo + step].normal_(generator=g)
    s1 = torch.randn(int(off1[-1]), dtype=torch.float32, device=dev, generator=g) + 0.1
    o0 = torch.from_numpy(off0).to(dev); o1 = torch.from_numpy(off1).to(dev)
    rid = torch.as_tensor((np.arange(P) // 500_009).astype(np.int32), device=dev)
    det = nm.DeviceDetector(0, nb=nb, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL if tests_all else L.TEST_KS)
    full = {k: v.clone() for k, v in det.run(s0, s1, rid, off0=o0, off1=o1, max_n0=4000, max_n1=400).items()}
    torch.cuda.synchronize()
    assert int(full['status'].max().item()) == 0
    names = ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p') if tests_all else ('ks_d', 'ks_p')
    for lo, cnt in ((0, 800), (P // 2, 800), (P - 800, 800)):
        a = s0[int(off0[lo]):int(off0[lo + cnt])].cpu().numpy(); b = s1[int(off1[lo]):int(off1[lo + cnt])].cpu().numpy()
        exp = oracle_c.detect_batch(a, off0[lo:lo + cnt + 1] - off0[lo], b, off1[lo:lo + cnt + 1] - off1[lo],
                                    rid[lo:lo + cnt].cpu().numpy(), nb, 2.0, 'stouffer', tests=7 if tests_all else 1, threads=0)
        _cmp_sample(full, exp, lo, cnt, nb, names)
    for i in np.argsort(n0)[-8:]:
        a = s0[int(off0[i]):int(off0[i + 1])].cpu().numpy(); b = s1[int(off1[i]):int(off1[i + 1])].cpu().numpy()
        e = oracle_c.detect_batch(a, np.array([0, len(a)]), b, np.array([0, len(b)]), np.zeros(1, np.int32), 0, 2.0, 'ks',
                                  tests=7 if tests_all else 1, threads=1)
        assert abs(float(full['ks_d'][i].item()) - e['ks_d'][0]) <= 0.0
        if tests_all:
            assert float(full['mwu_u'][i].item()) == e['mwu_u'][0]
    # D, its p-value and U do not depend on which group is called the first
    swapped = det.run(s1, s0, rid, off0=o1, off1=o0, max_n0=400, max_n1=4000)
    torch.cuda.synchronize()
    assert torch.equal(swapped['ks_d'], full['ks_d']) and torch.equal(swapped['ks_p'], full['ks_p'])
    if tests_all:
        assert torch.equal(swapped['mwu_u'], full['mwu_u'])

