"""The simulation repeat loops of the reference's studies, array-native and batched on the device.

Reference: every repeat draws reads without replacement from a pool of modified ("case") and unmodified
("control") reads, rebuilds the per-position sample lists from the drawn reads' events
(mySimulate.getGenomeEvents, mySimulate.py:124-139), applies the coverage filter and `mtest2`
(mySimulat2.py:161-165, myDownSampling0.py:91,115-116) and records the rank of the known modified site among the
ranked positions (mySimulate.getTopRank, mySimulate.py:287-328).  mySimulat2.py:127-171 varies the number of
modified reads at a fixed mixing percentage; myDownSampling0.py:62-123 down-samples both groups.

Here a pool is a set of flat arrays (`ReadPool`), the events of a drawn read set are expanded, keyed by
(chrom, strand, position) and grouped by ONE device sort (`group_events`: torch is the plumbing — memory, sort,
prefix sums), the positions present in both groups with enough coverage are intersected on the device, and the
per-position tests + combine + ranking run through the C ABI (nanomod_amd.engine: the HIP kernels).  Only the
p-value tracks of a repeat come back to the host, where `get_top_rank` walks the ranking.

The reference draws from an unseeded global RNG; here every repeat has a seed.  With the draws given (the golden
fixtures of oracle/gen_golden_sim.py) the results are the reference's, rank for rank.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from . import detect, engine

TARGET_CHR, TARGET_STRAND, TARGET_POS = 'spel', '-', 3072            # mySimulate.py:26-28 (the reference hard-codes its site)


class ReadPool:
    """Reads as readEvents holds them (mySimulate.py:101-122), flattened: per read `chrom`, `strand`, `start`
    (mapped_start) and a row `off[r] .. off[r+1]` of the event arrays `norm_mean` (float64) and `base`."""

    def __init__(self, chrom, strand, start, off, norm_mean, base, device=None):
        import torch
        self.torch = torch
        self.device = torch.device(device if device is not None else 'cpu')
        names = sorted(set(zip([str(c) for c in chrom], [str(s) for s in strand])))          # (chrom, strand) tuples, sorted as mtest2 iterates
        self.cs_names = names
        ids = {cs: i for i, cs in enumerate(names)}
        self.n = len(start)
        self.cs = torch.as_tensor(np.array([ids[(str(c), str(s))] for c, s in zip(chrom, strand)], dtype=np.int64), device=self.device)
        self.minus = torch.as_tensor(np.array([str(s) == '-' for s in strand]), device=self.device)
        self.start = torch.as_tensor(np.asarray(start, dtype=np.int64), device=self.device)
        self.off = torch.as_tensor(np.asarray(off, dtype=np.int64), device=self.device)
        self.norm_mean = torch.as_tensor(np.asarray(norm_mean, dtype=np.float64), device=self.device)
        self.base = torch.as_tensor(np.frombuffer(''.join(str(b)[:1] or ' ' for b in base).encode('latin-1'), dtype=np.uint8).copy(), device=self.device)


def _expand(pool, sel, cs_map):
    """(key, value, base) of every event of the reads `sel` of `pool`; key = cs << 40 | position
    (getGenomeEvents: position = start + i on '+', start + len - 1 - i on '-', mySimulate.py:133-136)"""
    torch = pool.torch
    sel = torch.as_tensor(sel, dtype=torch.int64, device=pool.device)
    lens = pool.off[sel + 1] - pool.off[sel]
    total = int(lens.sum().item())
    first = torch.cumsum(lens, 0) - lens
    r = torch.repeat_interleave(torch.arange(len(sel), device=pool.device), lens, output_size=total)
    i = torch.arange(total, device=pool.device) - first[r]
    src = pool.off[sel][r] + i
    pos = torch.where(pool.minus[sel][r], pool.start[sel][r] + lens[r] - 1 - i, pool.start[sel][r] + i)
    key = (cs_map[pool.cs[sel][r]] << 40) | pos
    return key, pool.norm_mean[src], pool.base[src]


def group_events(parts, cs_names):
    """getGenomeEvents for one dataset label: `parts` = [(pool, selected read indices), ...] whose events all go to
    the same group.  Returns device tensors: key (sorted unique positions), off (CSR), sig (float64), base (uint8)."""
    torch = parts[0][0].torch
    dev = parts[0][0].device
    ids = {cs: i for i, cs in enumerate(cs_names)}
    keys, vals, bases = [], [], []
    for pool, sel in parts:
        cs_map = torch.as_tensor([ids[cs] for cs in pool.cs_names], dtype=torch.int64, device=dev)
        k, v, b = _expand(pool, sel, cs_map)
        keys.append(k); vals.append(v); bases.append(b)
    key = torch.cat(keys); val = torch.cat(vals); base = torch.cat(bases)
    # (on the device: the library's own radix sort, nmod_argsort_keys; a CPU pool — the tests' — sorts with torch)
    order = engine.argsort_device(key.contiguous()) if key.is_cuda else torch.argsort(key, stable=True)
    key = key[order]
    ukey, counts = torch.unique_consecutive(key, return_counts=True)
    off = torch.zeros(len(ukey) + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(counts, 0)
    # (the reference overwrites base[(chrom, strand)][pos] with every event: the LAST one in read / event order stays)
    return {'key': ukey, 'off': off, 'sig': val[order], 'base': base[order][off[1:] - 1]}


def tested_positions(g0, g1, min_coverage):
    """mfilter_coverage (per group) + the intersection in mtest2's order (myDetect.py:301-314,421-431), on the device:
    CSR rows of both groups for the tested positions."""
    import torch
    out = []
    keep = []
    for g in (g0, g1):
        n = g['off'][1:] - g['off'][:-1]
        keep.append(torch.nonzero(n >= min_coverage).squeeze(1))
    k0, k1 = g0['key'][keep[0]], g1['key'][keep[1]]
    in1 = torch.isin(k0, k1, assume_unique=True)
    in0 = torch.isin(k1, k0, assume_unique=True)
    rows0, rows1 = keep[0][in1], keep[1][in0]                      # both ascending in key: aligned
    for g, rows in ((g0, rows0), (g1, rows1)):
        lens = g['off'][rows + 1] - g['off'][rows]
        noff = torch.zeros(len(rows) + 1, dtype=torch.int64, device=lens.device)
        noff[1:] = torch.cumsum(lens, 0)
        total = int(noff[-1].item())
        r = torch.repeat_interleave(torch.arange(len(rows), device=lens.device), lens, output_size=total)
        idx = g['off'][rows][r] + (torch.arange(total, device=lens.device) - noff[:-1][r])
        out.append((g['sig'][idx].contiguous(), noff))
    key = g0['key'][rows0]
    return key, g1['base'][rows1], out[0], out[1]


def get_top_rank(chrom, strand, pos, order, run_id, nb, window, region_rank=False,
                 target=(TARGET_CHR, TARGET_STRAND, TARGET_POS)):
    """mySimulate.getTopRank (mySimulate.py:287-328) on array-shaped records: walk the ranking `order`, skip records
    closer than `closesize` to an accepted one on the same (chrom, strand), accept those whose +-window neighbours
    are all consecutive positions of the same run (pos_check), and return the 1-based count of accepted records
    when one lies within closesize of the target site (-1 if the ranking ends first)."""
    closesize = nb * 2
    if region_rank:
        closesize = max(window, 1)
    t_chr, t_strand, t_pos = target
    best_pos = -t_pos
    if best_pos > -1000000:
        best_pos = -1000000
    n = len(pos)
    blocked = {}                                       # (chrom, strand) -> positions closer than closesize to an accepted one
    curn = 0                                           # (the reference compares with every accepted record: quadratic)
    for i in order:
        i = int(i)
        cs = (chrom[i], strand[i])
        p = int(pos[i])
        near = blocked.get(cs)
        if near is not None and p in near:
            continue
        lo, hi = i - window, i + window
        if lo < 0 or hi > n - 1 or run_id[lo] != run_id[i] or run_id[hi] != run_id[i]:
            continue                                   # some neighbour is not a consecutive position: not enough
        if near is None:
            near = blocked[cs] = set()
        near.update(range(p - closesize + 1, p + closesize))
        curn += 1
        if cs == (t_chr, t_strand) and abs(best_pos - t_pos) > abs(p - t_pos) and abs(p - t_pos) < closesize:
            return curn
    return -1


def run_repeat(case_parts, control_parts, opts, device=0):
    """One trip of the reference's repeat loops with the draws given: `case_parts` / `control_parts` are lists of
    (ReadPool, selected read indices) feeding dataset 1 ('simulate_case') and dataset 2 ('folder_control').
    opts: 'MinCoverage', 'neighborPvalues', 'WeightsDif', 'testMethod', 'rankUse', 'window', 'RegionRankbyST', and —
    as the reference's loops, which call mtest2 itself — 'coverages' (+ 'downsampling', 'downsampling_quantile',
    'seed'): the down-sampling branch of getKStest (myDetect.py:345-361) through detect.downsample_update.
    The device is the pools'; `device` is used only for host-resident pools.
    Returns (rank of the target site, dict of the tested positions and their numbers)."""
    import torch
    cs_names = sorted(set(cs for pool, _ in case_parts + control_parts for cs in pool.cs_names))
    g0 = group_events(case_parts, cs_names)
    g1 = group_events(control_parts, cs_names)
    key, base, (sig0, off0), (sig1, off1) = tested_positions(g0, g1, opts['MinCoverage'])
    npos = len(key)
    cs = (key >> 40).cpu().numpy()
    pos = (key & ((1 << 40) - 1)).cpu().numpy()
    chrom = np.array([cs_names[c][0] for c in cs], dtype=object)
    strand = np.array([cs_names[c][1] for c in cs], dtype=object)
    rid = detect.run_ids(chrom, strand, pos)
    method, nb = opts['testMethod'], opts['neighborPvalues']
    dev_method = method if (method in ('stouffer', 'fisher') and nb > 0) else 'ks'
    if npos == 0:
        return -1, dict(chrom=chrom, strand=strand, pos=pos)
    if sig0.is_cuda:
        device = sig0.device.index or 0
        det = engine.DeviceDetector(device, nb=max(nb, 0), weights_dif=opts.get('WeightsDif', 2.0), method=dev_method)
        n0 = (off0[1:] - off0[:-1]); n1 = (off1[1:] - off1[:-1])
        r = det.run(sig0, sig1, torch.as_tensor(rid, device=sig0.device), off0=off0, off1=off1,
                    max_n0=int(n0.max().item()), max_n1=int(n1.max().item()))
        res = {k: v.cpu().numpy() for k, v in r.items()}
    else:                                              # host tensors: the library stages them (still the HIP path)
        res = engine.detect_host(sig0.numpy(), off0.numpy(), sig1.numpy(), off1.numpy(), rid, nb=max(nb, 0),
                                 weights_dif=opts.get('WeightsDif', 2.0), method=dev_method, device=device)
    if np.any(res['status'] & L.STATUS_MWU_ALL_IDENTICAL):
        raise ValueError('All numbers are identical in mannwhitneyu')
    cov = opts.get('coverages')
    if cov is not None and (int(cov[0]) > 0 or int(cov[1]) > 0):
        detect.downsample_update(res, sig0.cpu().numpy(), off0.cpu().numpy(), sig1.cpu().numpy(), off1.cpu().numpy(), rid, strand, cov,
                                 iters=opts.get('downsampling', 100), quantile=opts.get('downsampling_quantile', 0.25),
                                 seed=opts.get('seed', 0), nb=max(nb, 0), weights_dif=opts.get('WeightsDif', 2.0),
                                 method=dev_method, device=device)
    use_p = opts.get('rankUse', 'pv') == 'pv'
    ks_key = res['ks_p'] if use_p else res['ks_d']
    mw_key = res['mwu_p'] if use_p else res['mwu_u']
    first = ks_key if (method == 'ks' or nb == 0) else res['comb_p' if use_p else 'comb_st']
    order = engine.rank_order_host(first, ks_key, mw_key, descending=not use_p, device=device)
    rank = get_top_rank(chrom, strand, pos, order, rid, nb, opts['window'], opts.get('RegionRankbyST', 0) == 1)
    tab = dict(chrom=chrom, strand=strand, pos=pos, base=base.cpu().numpy(), n0=(off0[1:] - off0[:-1]).cpu().numpy(),
               n1=(off1[1:] - off1[:-1]).cpu().numpy(), order=order, **res)
    return rank, tab


def _draw(torch, n, k, gen, device):
    return torch.randperm(n, generator=gen, device=device)[:k]          # np.random.choice(n, k, replace=False)


def simulate_case_size(case_pool, control_pool, case_size, percentage, random_times, opts, seed=0):
    """mySimulat2.mSimulate1's loop (mySimulat2.py:127-171): `case_size` modified reads mixed with unmodified ones so
    that they make up `percentage` of dataset 1, against int(case_size / percentage) other unmodified reads.
    Returns the list of ranks ('PercDis')."""
    torch = case_pool.torch
    gen = torch.Generator(device=case_pool.device); gen.manual_seed(int(seed))
    n_un1 = int(case_size * (1 - percentage) / percentage)
    n_un2 = int(case_size / percentage)
    ranks = []
    for _ in range(random_times):
        c = _draw(torch, case_pool.n, case_size, gen, case_pool.device)
        k = _draw(torch, control_pool.n, n_un1 + n_un2, gen, case_pool.device)
        rank, _ = run_repeat([(case_pool, c), (control_pool, k[:n_un1])], [(control_pool, k[n_un1:])], opts)
        ranks.append(rank)
    return ranks


def down_sampling(case_pool, control_pool, case_size, random_times, opts, seed=0, max_draws=None,
                  target=(TARGET_CHR, TARGET_STRAND, TARGET_POS)):
    """myDownSampling0.mSimulate1's loop (myDownSampling0.py:62-123): both groups down-sampled to `case_size` reads
    (up to 30 % more after repeated shallow draws); a draw is rejected when more than two of the 14 (group, position)
    pairs around the target hold fewer than 0.95 * case_size / 5 samples.  Returns the list of ranks ('CovgDis')."""
    torch = case_pool.torch
    gen = torch.Generator(device=case_pool.device); gen.manual_seed(int(seed))
    cs_names = sorted(set(case_pool.cs_names) | set(control_pool.cs_names))
    t_cs = cs_names.index((target[0], target[1])) if (target[0], target[1]) in cs_names else -1
    ranks = []
    rt = repeat_time = cur_repeat_time = draws = 0
    while rt < random_times and (max_draws is None or draws < max_draws):
        draws += 1
        more = min(repeat_time, 15)
        k = int(case_size * (1 + more * 0.02))
        c = _draw(torch, case_pool.n, k, gen, case_pool.device) if case_pool.n > k else torch.arange(case_pool.n, device=case_pool.device)
        d = _draw(torch, control_pool.n, k, gen, case_pool.device) if control_pool.n > k else torch.arange(control_pool.n, device=case_pool.device)
        shallow = 0
        for parts in ([(case_pool, c)], [(control_pool, d)]):
            g = group_events(parts, cs_names)
            cov = dict(zip(g['key'].cpu().tolist(), (g['off'][1:] - g['off'][:-1]).cpu().tolist()))
            if t_cs >= 0 and any((g['key'] >> 40 == t_cs).cpu().tolist()):
                for p in range(target[2] - 3, target[2] + 4):
                    if cov.get((t_cs << 40) | p, 0) < 0.95 * case_size / 5:
                        shallow += 1
        if shallow > 2:
            if shallow > 3 and cur_repeat_time > 5:
                repeat_time += 1
            cur_repeat_time += 1
            continue
        rank, _ = run_repeat([(case_pool, c)], [(control_pool, d)], opts)
        ranks.append(rank)
        rt += 1
        cur_repeat_time = 0
    return ranks
