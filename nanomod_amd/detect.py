"""Host-side mirror of the reference's operator interface for the hot path.

Same names, argument meaning and error behaviour as
/root/reference/bin/scripts/myDetect.py:301-545 — `mfilter_coverage`,
`m_min_float`, `m_max_float`, `getKStest`, `pos_check`, `combin_pvalues`,
`get_combin_pvalue`, `mtest2`, `save_test` — so a caller holding the reference's
`moptions` dict can switch modules.  The arithmetic runs in the HIP library
(nanomod_amd/libnanomod_hip.so) through the C ABI; there is no CPU fallback.

`moptions` keys read (as the reference): 'ds2', <dataset>['norm_mean'|'base'|
'basedict'][(chrom,strand)][pos], 'MinCoverage', 'neighborPvalues',
'WeightsDif', 'testMethod', 'rankUse', 'SaveTest', 'outFolder', 'FileID',
'mstd', 'coverages', 'RegionRankbyST' (+ 'window', 'WindOvlp', 'percentile', 'NA'), 'outLevel'.
Keys written: 'sign_test', 'sorted_sign_test', optionally 'sign_test_mstd', 'nmod_flagged' (positions flagged TOO_LARGE / NONFINITE; 'nmod_strict' raises),
plus 'sign_test_arrays' (the same numbers as numpy arrays, an addition).
"""
from __future__ import annotations

import collections.abc
import sys
import time

import numpy as np

from . import _lib as L
from . import engine

OUTPUT_DEBUG, OUTPUT_INFO, OUTPUT_WARNING, OUTPUT_ERROR = 0, 1, 2, 3   # myCom.py:5-8


# myDetect.py:317-325
def m_min_float(fv):
    if fv < sys.float_info.min:
        return sys.float_info.min
    return fv


def m_max_float(fv):
    if fv > sys.float_info.max:
        return sys.float_info.max
    return fv


def _hostwalk_module():
    """the C walk over the reference's dicts (csrc/hostwalk.c); None when it was not built"""
    try:
        from . import _hostwalk
        return _hostwalk
    except ImportError:
        return None


# myDetect.py:301-314
def mfilter_coverage(moptions):
    hw = _hostwalk_module()
    for dsn in moptions['ds2']:
        curds = moptions[dsn]['norm_mean']
        for sk in sorted(curds.keys()):
            bases = moptions[dsn]['base'].get(sk) if type(moptions[dsn]['base']) is dict else None
            if (hw is not None and type(curds[sk]) is dict and type(bases) is dict and
                    isinstance(moptions['MinCoverage'], (int, np.integer)) and not isinstance(moptions['MinCoverage'], bool)):
                # one C pass over the strand's dict in storage order (4.6 M positions: 0.05 s instead of 0.5 - 1.0 s).  Only for an
                # integer threshold (len(row) < 4.5 keeps 5, an int() of it would not) and when the strand has its base dict — the
                # loop below raises for a missing one only when it actually deletes, like the reference
                hw.filter_coverage(curds[sk], bases, int(moptions['MinCoverage']))
            else:
                for pk in sorted(curds[sk].keys()):
                    if len(curds[sk][pk]) < moptions['MinCoverage']:
                        del curds[sk][pk]
                        del moptions[dsn]['base'][sk][pk]
            if len(curds[sk]) == 0:
                del curds[sk]
                del moptions[dsn]['base'][sk]


# ---------------------------------------------------------------------------
# Above this many samples the host does not look for a narrower dtype: the float64 values go to the device as they are
# and the front end of NMOD_DTYPE_F64 picks order-preserving float32 keys per position there (nanomod_hip.hip:
# detect_f64).  The host-side search is ~8 passes over the vector (1.2 s of the 3.6 s of a 4.6 M-position mtest2).
DEVICE_ENCODE_ABOVE = 4_000_000


def encode_signals(values):
    """Pick the device dtype for a float64 sample vector without changing a
    single value the reference would see: float32 if every value is
    float32-exact, else int16 milli-units if every value is k/1000.0 (NanoMod's
    Events are 3-dp rounded, myRefBaseSignalAnnotation.py:1108), else the
    float64 values themselves (NMOD_DTYPE_F64: the device picks keys per position)."""
    v = np.asarray(values, dtype=np.float64)
    f32 = v.astype(np.float32)
    if np.array_equal(f32.astype(np.float64), v):
        return f32
    k = np.rint(v * 1000.0)
    if np.all(np.abs(k) <= 32767) and np.array_equal(k / 1000.0, v):
        return k.astype(np.int16)
    return v


def encode_pair(sig0, sig1):
    """encode_signals for the two groups of a batch (one dtype for both); large float64 batches pass through"""
    if sig0.dtype == np.float64 and sig1.dtype == np.float64 and sig0.size + sig1.size > DEVICE_ENCODE_ABOVE:
        return np.ascontiguousarray(sig0), np.ascontiguousarray(sig1)
    both = encode_signals(np.concatenate([sig0, sig1]))
    return both[:len(sig0)], both[len(sig0):]


def _coverage_threshold(moptions, m_str):
    cov = moptions.get('coverages', (0, 0))
    return int(cov[0 if m_str == '+' else 1])


# myDetect.py:327-363
def getKStest(moptions, a, b, m_str):
    cov = _coverage_threshold(moptions, m_str)
    both = encode_signals(np.concatenate([np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)]))
    sig0, sig1 = both[:len(a)], both[len(a):]
    off0 = np.array([0, len(a)], dtype=np.int64)
    off1 = np.array([0, len(b)], dtype=np.int64)
    r = engine.detect_host(sig0, off0, sig1, off1, np.zeros(1, np.int32), method='ks',
                           device=moptions.get('nmod_device', 0))
    if r['status'][0] & L.STATUS_MWU_ALL_IDENTICAL:
        raise ValueError('All numbers are identical in mannwhitneyu')      # scipy 1.2.1 behaviour
    ks = (float(r['ks_d'][0]), float(r['ks_p'][0]))
    if not (cov <= 0 or (len(a) <= cov and len(b) <= cov)):                  # myDetect.py:345-361, seeded
        dsd, dsp = engine.downsample_ks(sig0, off0, sig1, off1, [0], [cov], iters=int(moptions.get('downsampling', 100)),
                                        quantile=float(moptions.get('downsampling_quantile', 0.25)),
                                        seed=int(moptions.get('nmod_seed', 0)), device=moptions.get('nmod_device', 0))
        ks = (float(dsd[0]), float(dsp[0]))
    return [(float(r['mwu_u'][0]), float(r['mwu_p'][0])), (float(r['t_t'][0]), float(r['t_p'][0])), ks]


# myDetect.py:366-371
def pos_check(mlist, i, j):
    if j < 0 or j > len(mlist) - 1:
        return False
    if i == j or (mlist[i][0][0] == mlist[j][0][0] and mlist[i][0][1] == mlist[j][0][1]
                  and i - j == mlist[i][0][2] - mlist[j][0][2]):
        return True
    return False


def run_ids(chroms, strands, positions):
    """run_id[i] == run_id[j]  <=>  pos_check holds for every pair between i and j:
    same chrom, same strand, consecutive positions."""
    n = len(positions)
    rid = np.zeros(n, dtype=np.int32)
    if n == 0:
        return rid
    pos = np.asarray(positions, dtype=np.int64)
    brk = np.ones(n, dtype=bool)
    same = (pos[1:] - pos[:-1] == 1)
    chroms = np.asarray(chroms)
    strands = np.asarray(strands)
    same &= (chroms[1:] == chroms[:-1]) & (strands[1:] == strands[:-1])
    brk[1:] = ~same
    return (np.cumsum(brk) - 1).astype(np.int32)


def _sign_test_track(moptions):
    st = moptions['sign_test']
    ks_d = np.array([r[1][2][0] for r in st], dtype=np.float64)
    ks_p = np.array([r[1][2][1] for r in st], dtype=np.float64)
    rid = run_ids([r[0][0] for r in st], [r[0][1] for r in st], [r[0][2] for r in st])
    return ks_d, ks_p, rid


# myDetect.py:379-414
def get_combin_pvalue(moptions, i):
    if moptions['neighborPvalues'] > 0 and len(moptions['sign_test']) > 0:
        if moptions['testMethod'] not in ('fisher', 'stouffer'):
            raise UnboundLocalError("local variable 'comb_p_p' referenced before assignment")   # as the reference
        nb = moptions['neighborPvalues']
        st = moptions['sign_test']
        # window = [p_KS[j] if usable else 1.0]  (myDetect.py:383-389), sent as one tiny track
        win = [st[j][1][2][1] if (0 <= j <= len(st) - 1 and pos_check(st, i, j)) else 1.0
               for j in range(i - nb, i + nb + 1)]
        cst, cp = engine.combine_host(np.zeros(len(win)), np.array(win), np.zeros(len(win), np.int32), nb=nb,
                                      weights_dif=moptions['WeightsDif'], method=moptions['testMethod'],
                                      device=moptions.get('nmod_device', 0))
        return (float(cst[nb]), float(cp[nb]))
    if moptions['neighborPvalues'] == 0:
        return moptions['sign_test'][i][1][2]
    return None


# myDetect.py:373-377 (whole track in one launch instead of one call per position)
def combin_pvalues(moptions):
    st = moptions['sign_test']
    if len(st) == 0:
        return
    nb = moptions['neighborPvalues']
    if nb == 0:
        for r in st:
            r[1].append(r[1][2])
        return
    if nb < 0:
        return
    ks_d, ks_p, rid = _sign_test_track(moptions)
    cst, cp = engine.combine_host(ks_d, ks_p, rid, nb=nb, weights_dif=moptions['WeightsDif'],
                                  method=moptions['testMethod'], device=moptions.get('nmod_device', 0))
    for r, s, p in zip(st, cst.tolist(), cp.tolist()):
        r[1].append((s, p))


# ---------------------------------------------------------------------------
class SignTestRecords(collections.abc.Sequence):
    """`moptions['sign_test']` as the reference builds it (myDetect.py:436) — a sequence of
    ((chrom, strand, pos, base, n0, n1), [(U, pU), (t, pt), (D, pKS)[, (comb stat, comb p)]]) records — backed by
    the result arrays: a record is built when it is first asked for and kept (so `sorted_sign_test` holds the same
    objects), 4.6 M tuples are not built up front.  A `collections.abc.Sequence` (no item assignment, no append; a record itself can be
    edited in place and the edit is kept — `edited()`): len, indexing, slicing,
    iteration, sorted(), `in`, reversed(), `.count()`, `.index()` (what mySimulate.getTopRank calls on it,
    mySimulate.py:312 — constant time for a record obtained from this object or its ranked view), `==` against a
    list of records, `+` (gives a list); it pickles as a plain list."""

    def __init__(self, meta, res, with_comb, order=None, parent=None):
        self._meta, self._res, self._with_comb = meta, res, with_comb
        self._order = order
        self._parent = parent
        self._cache = {} if parent is None else None
        self._where = {} if parent is None else None     # id(record) -> position index, for index()
        self._inv = None

    def __len__(self):
        return len(self._order) if self._order is not None else len(self._meta['pos'])

    def _build(self, i):
        m, r = self._meta, self._res
        rec = ((str(m['chrom'][i]), str(m['strand'][i]), int(m['pos'][i]), str(m['base'][i]), int(m['n0'][i]), int(m['n1'][i])),
               [(float(r['mwu_u'][i]), float(r['mwu_p'][i])), (float(r['t_t'][i]), float(r['t_p'][i])),
                (float(r['ks_d'][i]), float(r['ks_p'][i]))])
        if self._with_comb:
            rec[1].append((float(r['comb_st'][i]), float(r['comb_p'][i])))
        return rec

    def _get(self, i):
        if self._parent is not None:
            return self._parent._get(i)
        rec = self._cache.get(i)
        if rec is None:
            rec = self._cache[i] = self._build(i)
            self._where[id(rec)] = i
        return rec

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        n = len(self)
        if k < 0:
            k += n
        if not 0 <= k < n:
            raise IndexError('list index out of range')
        return self._get(int(self._order[k]) if self._order is not None else k)

    def _build_block(self, idx):
        """records for the position indices `idx` (an int array), built column-wise: one `tolist()` per column instead of
        fourteen scalar conversions per record (0.5 us against 4 us a record when a consumer walks the whole list)"""
        root = self._parent if self._parent is not None else self
        m, r = self._meta, self._res
        cols = [np.asarray(m[k])[idx].tolist() for k in ('chrom', 'strand', 'pos', 'base', 'n0', 'n1')]
        cols[0] = [str(x) for x in cols[0]]; cols[1] = [str(x) for x in cols[1]]; cols[3] = [str(x) for x in cols[3]]
        nums = [np.asarray(r[k], dtype=np.float64)[idx].tolist() for k in ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p')]
        comb = [np.asarray(r[k], dtype=np.float64)[idx].tolist() for k in ('comb_st', 'comb_p')] if self._with_comb else None
        out = []
        for j, i in enumerate(idx.tolist()):
            rec = root._cache.get(i)
            if rec is None:
                tests = [(nums[0][j], nums[1][j]), (nums[2][j], nums[3][j]), (nums[4][j], nums[5][j])]
                if comb is not None:
                    tests.append((comb[0][j], comb[1][j]))
                rec = ((cols[0][j], cols[1][j], cols[2][j], cols[3][j], cols[4][j], cols[5][j]), tests)
                root._cache[i] = rec
                root._where[id(rec)] = i
            out.append(rec)
        return out

    def __iter__(self):
        n = len(self)
        block = 4096
        for lo in range(0, n, block):
            hi = min(lo + block, n)
            idx = np.asarray(self._order[lo:hi], dtype=np.int64) if self._order is not None else np.arange(lo, hi, dtype=np.int64)
            for rec in self._build_block(idx):
                yield rec

    def index(self, value, start=0, stop=None):
        """list.index: first k with self[k] == value.  A record handed out by this object (or by the view that shares its
        records) is found through its identity; anything else by the equality scan of a list."""
        root = self._parent if self._parent is not None else self
        i = root._where.get(id(value))
        if i is not None and root._cache.get(i) is value:
            if self._order is None:
                k = i
            else:
                if self._inv is None:
                    self._inv = {int(p): kk for kk, p in enumerate(np.asarray(self._order).tolist())}
                k = self._inv.get(i)
            n = len(self)
            lo = max(start + n, 0) if start < 0 else start
            hi = n if stop is None else (max(stop + n, 0) if stop < 0 else min(stop, n))
            if k is not None and lo <= k < hi:
                return k
            raise ValueError('%r is not in list' % (value,))
        return super().index(value, start, len(self) if stop is None else stop)

    def __eq__(self, other):
        if isinstance(other, (list, SignTestRecords)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    __hash__ = None

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def tolist(self):
        return list(self)

    def __reduce__(self):
        return (list, (list(self),))

    def __repr__(self):
        return 'SignTestRecords(%d records%s)' % (len(self), ', ranked view' if self._order is not None else '')

    def edited(self):
        """True when a record handed out by this object no longer equals what the result arrays hold — a caller edited it in
        place the reference's way (`sign_test[i][1].append(...)`, myDetect.py:377: the records are kept once built, so such an
        edit persists in the sequence).  save_test then writes the table from the records, not from the arrays."""
        root = self._parent if self._parent is not None else self
        return any(not _same_record(rec, root._build(i)) for i, rec in root._cache.items())

    def permuted(self, order):
        """the same records in another order (the ranking)"""
        return SignTestRecords(self._meta, self._res, self._with_comb, order=np.asarray(order), parent=self if self._parent is None else self._parent)


def _same_record(a, b):
    """record equality in which a NaN equals a NaN: a position with a NaN statistic (zero variance, a flagged position) rebuilt from the
    arrays holds a different float object each time, and `!=` on the tuples would call an untouched record edited"""
    if a[0] != b[0] or len(a[1]) != len(b[1]):
        return False
    for x, y in zip(a[1], b[1]):
        try:
            if len(x) != len(y):
                return False
            for u, v in zip(x, y):
                if not (u == v or (u != u and v != v)):
                    return False
        except TypeError:                                     # (an edit that put something else than a (stat, p) pair there)
            return False
    return True


def _join_strand_py(d0, d1, b0, b1, sk, quiet):
    """the positions of both datasets of one (chrom, strand) in ascending order and their rows as CSR pieces, without the C
    module (or for position keys it does not take): set intersection, sort, itemgetter and map iterate in C"""
    import operator
    common = sorted(d0.keys() & d1.keys())
    if not common:
        z = np.zeros(0)
        return np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), z, z, []
    fetch = (lambda d: (d[common[0]],)) if len(common) == 1 else operator.itemgetter(*common)
    a_rows, b_rows = fetch(d0), fetch(d1)
    b0v, b1v = fetch(b0), fetch(b1)
    if b0v != b1v and not quiet:
        for pk, x1, x0 in zip(common, b1v, b0v):
            if not x1 == x0:
                print('Error not equal', sk, pk, x1, x0)
    n0 = np.fromiter(map(len, a_rows), dtype=np.int32, count=len(common))
    n1 = np.fromiter(map(len, b_rows), dtype=np.int32, count=len(common))

    def flat(chunks, total):
        hw = _hostwalk_module()
        if hw is not None:
            out = np.empty(total, dtype=np.float64)
            if hw.flatten(chunks, out) != total:
                raise ValueError('build_csr: a position changed its number of samples while it was read')
            return out
        if isinstance(chunks[0], np.ndarray):
            return np.concatenate(chunks).astype(np.float64, copy=False)
        import itertools
        return np.fromiter(itertools.chain.from_iterable(chunks), dtype=np.float64, count=total)
    return (np.array(common, dtype=np.int64), n0, n1, flat(a_rows, int(n0.sum(dtype=np.int64))), flat(b_rows, int(n1.sum(dtype=np.int64))),
            list(b1v))


def _gather_rows(hw, parts, plans, total0, total1):
    """The rows of every (chrom, strand) in ONE pair of CSR arrays, in the device dtype encode_signals would pick.  Planned strands
    (csrc/hostwalk.c: join_strand_plan) are copied by the C walk's threads straight into the arrays — first as int16 milli-units,
    narrowed on the way (what stored events are: x = k / 1000.0; 2 bytes written per 8 read), and only when a sample refuses that
    as float64, which then goes through encode_pair as before.  A strand that took the Python walk brings its own arrays."""
    if not all(p is not None for p in plans):             # a strand of exotic rows / keys: the general way (per-strand arrays, one concatenate)
        for i, (part, plan) in enumerate(zip(parts, plans)):
            if plan is not None:
                s0 = np.empty(int(part[1].sum(dtype=np.int64)), dtype=np.float64); s1 = np.empty(int(part[2].sum(dtype=np.int64)), dtype=np.float64)
                hw.copy_plan(plan, s0, 0, s1, 0)
                parts[i] = part[:3] + (s0, s1) + part[5:]
        cat = lambda k: np.concatenate([p[k] for p in parts]) if len(parts) > 1 else parts[0][k]
        return encode_pair(cat(3), cat(4))
    for dt in (np.int16, np.float64):
        sig0 = np.empty(total0, dtype=dt); sig1 = np.empty(total1, dtype=dt)
        at0 = at1 = 0
        ok = True
        for part, plan in zip(parts, plans):
            if not hw.copy_plan(plan, sig0, at0, sig1, at1):
                ok = False
                break
            at0 += int(part[1].sum(dtype=np.int64)); at1 += int(part[2].sum(dtype=np.int64))
        if ok:
            return (sig0, sig1) if dt == np.int16 else encode_pair(sig0, sig1)
    raise AssertionError('copy_plan refused float64 output')


def build_csr(moptions):
    """The tested-position set and order of mtest2 (myDetect.py:421,427-431) as CSR arrays + array-shaped metadata.

    Three input shapes, fastest first: (i) both datasets carry `'nmod_container'` — the flat arrays of
    nanomod_amd.container (what a loader that never builds the dicts attaches): filter, intersection and order are
    numpy operations (cli.select_positions; the caller has NOT run mfilter_coverage, it happens here); (ii) the
    reference's dicts whose per-position values are numpy arrays: no per-value conversion; (iii) the reference's
    dicts of Python lists of numpy.float64 (myDetect.py:124)."""
    ds0 = moptions[moptions['ds2'][0]]
    ds1 = moptions[moptions['ds2'][1]]
    if 'nmod_container' in ds0 and 'nmod_container' in ds1:
        from . import cli
        meta, sig0, off0, sig1, off1, rid = cli.select_positions(ds0['nmod_container'], ds1['nmod_container'],
                                                                 moptions['MinCoverage'], moptions.get('outLevel', OUTPUT_ERROR))
        return meta, sig0, off0, sig1, off1, rid
    quiet = moptions.get('outLevel', OUTPUT_ERROR) > OUTPUT_ERROR
    hw = _hostwalk_module()
    chrom, strand, counts = [], [], []                    # one entry per (chrom, strand) key
    parts = []                                            # per key: (pos, n0, n1, sig0, sig1, bases) — sig0 / sig1 None for a planned strand
    plans = []                                            # per key: the C walk's plan (rows copied later, all strands into ONE pair of arrays) or None
    two_step = hw is not None and hasattr(hw, 'join_strand_plan')
    for sk in sorted(ds0['norm_mean'].keys()):
        if sk not in ds1['norm_mean']:
            continue
        d0, d1 = ds0['norm_mean'][sk], ds1['norm_mean'][sk]
        b0, b1 = ds0['base'][sk], ds1['base'][sk]
        part, plan = None, None
        if hw is not None and all(type(x) is dict for x in (d0, d1, b0, b1)):
            try:
                # the loop header of mtest2 for this strand in one C pass: merge join of the two dicts read in storage
                # order (csrc/hostwalk.c: join_strand_plan); the rows follow below, every strand's straight into the batch's CSR arrays
                if two_step:
                    p_, n0_, n1_, plan, bases_, mism, codes_ = hw.join_strand_plan(d0, d1, b0, b1)
                    s0_ = s1_ = None
                else:
                    p_, n0_, n1_, s0_, s1_, bases_, mism, codes_ = hw.join_strand(d0, d1, b0, b1)
                if mism and not quiet:
                    b0l = [b0[int(p_[j])] for j in mism]
                    for j, x0 in zip(mism, b0l):
                        print('Error not equal', sk, int(p_[j]), bases_[j], x0)
                part = (p_, n0_, n1_, s0_, s1_, bases_, codes_)
            except TypeError:                             # position keys that are not integers: the general way below
                part, plan = None, None
        if part is None:
            part = _join_strand_py(d0, d1, b0, b1, sk, quiet)
        if len(part[0]) == 0:
            continue
        chrom.append(sk[0]); strand.append(sk[1]); counts.append(len(part[0]))
        parts.append(part); plans.append(plan)
    cat = lambda k, dt: (np.concatenate([p[k] for p in parts]) if len(parts) > 1 else parts[0][k]) if parts else np.zeros(0, dtype=dt)
    pos = cat(0, np.int64); n0 = cat(1, np.int32); n1 = cat(2, np.int32)
    base = [x for p in parts for x in p[5]] if len(parts) != 1 else parts[0][5]
    npos = len(pos)
    off0 = np.zeros(npos + 1, dtype=np.int64)
    off1 = np.zeros(npos + 1, dtype=np.int64)
    if npos:
        np.cumsum(n0, out=off0[1:])
        np.cumsum(n1, out=off1[1:])
        sig0, sig1 = _gather_rows(hw, parts, plans, int(off0[-1]), int(off1[-1]))
    else:
        sig0 = sig1 = np.zeros(0, dtype=np.float32)
    names = sorted(set(chrom))
    ids = {c: i for i, c in enumerate(names)}
    counts = np.asarray(counts, dtype=np.int64)
    codes = [p[6] if len(p) > 6 else None for p in parts]
    if parts and all(c is not None and (len(c) == 0 or int(c.min()) > 0) for c in codes):
        # every base is a one-character string (what the reference stores): the code points ARE a '<U1' array, no copy
        base_arr = (np.concatenate(codes) if len(codes) > 1 else codes[0]).view('<U1')
    else:
        base_arr = np.empty(npos, dtype=object)
        base_arr[:] = base
    as_text = lambda names_: np.array(names_) if all(isinstance(x, str) for x in names_) else np.array(names_, dtype=object)
    meta = dict(chrom=np.repeat(as_text(chrom), counts), strand=np.repeat(as_text(strand), counts),
                pos=np.asarray(pos, dtype=np.int64), base=base_arr, n0=np.asarray(n0, dtype=np.int32),
                n1=np.asarray(n1, dtype=np.int32), names=names,
                chrom_id=np.repeat(np.array([ids[c] for c in chrom], dtype=np.int32), counts))
    rid = run_ids(meta['chrom'], meta['strand'], meta['pos'])
    return meta, sig0, off0, sig1, off1, rid


# myDetect.py:463-515 — ranking of windows instead of single positions (--RegionRankbyST 1)
def region_rank(moptions, sorted_ind, use_pind):
    """A window = the positions pk-w .. pk+w (w = moptions['window'] + 1: the reference increments the
    option in place) that all exist on one (chrom, strand) and lie below the last tested position of
    that strand; its key is the `percentile`-th smallest p (or statistic) of the window, ties broken by
    the distance of the window minimum from the centre.  Only bases equal to moptions['NA'] contribute
    when that option is set; windows with <= 5 contributing values are dropped.  With WindOvlp == 1 the
    windows slide by one position and a window is suppressed when a better-ranked one on the same strand
    lies closer than w.  The window keys and the ranking run on the device (nmod_region_rank); the records must
    be in the reference's order (sorted (chrom, strand), ascending position), which is how mtest2 builds them."""
    moptions['window'] = moptions['window'] + 1
    w = moptions['window']
    movesize = 1 if moptions['WindOvlp'] == 1 else w
    recs = moptions['sign_test']
    n = len(recs)
    if n == 0:
        return []
    keys = [(r[0][0], r[0][1]) for r in recs]
    pos = np.fromiter((r[0][2] for r in recs), dtype=np.int64, count=n)
    rank_of = {k: i for i, k in enumerate(sorted(set(keys)))}
    sid = np.fromiter((rank_of[k] for k in keys), dtype=np.int64, count=n)
    if np.any(np.diff(sid) < 0) or np.any((np.diff(sid) == 0) & (np.diff(pos) <= 0)):
        raise ValueError('region_rank: sign_test must be ordered by (chrom, strand) and ascending position')
    starts = np.flatnonzero(np.r_[True, np.diff(sid) != 0])
    ends = np.r_[starts[1:], n] - 1
    seg = np.searchsorted(starts, np.arange(n), side='right') - 1
    value = np.fromiter((r[1][sorted_ind][use_pind] for r in recs), dtype=np.float64, count=n)
    base = ''.join((r[0][3][:1] or ' ') for r in recs).encode('latin-1')
    idx = engine.region_rank_host(starts[seg], ends[seg], pos, base, value, w, movesize, moptions.get('NA', ''),
                                  moptions['percentile'], moptions['WindOvlp'], device=moptions.get('nmod_device', 0))
    return [recs[i] for i in idx.tolist()]


# myDetect.py:416-462
def downsample_update(res, sig0, off0, sig1, off1, rid, strands, coverages, *, iters=100, quantile=0.25, seed=0,
                      nb=2, weights_dif=2.0, method='stouffer', device=0):
    """Down-sampling branch (myDetect.py:339-361) on a finished batch: positions where a group exceeds its
    strand's coverage threshold get the KS pair of the `quantile`-th of `iters` resamples (seeded here,
    unseeded in the reference) and the combined track is recomputed; MWU / Welch keep the full data.
    Returns the indices of the positions that were down-sampled."""
    c = (int(coverages[0]), int(coverages[1]))
    if c[0] <= 0 and c[1] <= 0:
        return np.zeros(0, dtype=np.int64)
    cov_pos = np.where(np.asarray(strands) == '+', c[0], c[1]).astype(np.int64)
    n0a = np.diff(off0); n1a = np.diff(off1)
    flag = np.nonzero((cov_pos > 0) & ((n0a > cov_pos) | (n1a > cov_pos)))[0]
    if len(flag):
        dsd, dsp = engine.downsample_ks(sig0, off0, sig1, off1, flag, cov_pos[flag], iters=iters, quantile=quantile,
                                        seed=seed, device=device)
        res['ks_d'][flag] = dsd
        res['ks_p'][flag] = dsp
        if method != 'ks':
            res['comb_st'], res['comb_p'] = engine.combine_host(res['ks_d'], res['ks_p'], rid, nb=nb, weights_dif=weights_dif,
                                                                method=method, device=device)
    return flag


def mtest2(moptions):
    print("Start sorting")
    engine.warm_up(moptions.get('nmod_device', 0))          # the HIP start-up runs beside the host-side preparation
    meta, sig0, off0, sig1, off1, rid = build_csr(moptions)
    npos = len(meta['pos'])
    method = moptions['testMethod']
    nb = moptions['neighborPvalues']
    want_mstd = not moptions.get('mstd', 0) == 0
    dev = moptions.get('nmod_device', 0)
    start_time = time.time()
    # the combine is skipped for 'ks' (myDetect.py:443); for nb == 0 it returns the KS tuple (:413)
    dev_method = method if (method in ('stouffer', 'fisher') and nb > 0) else 'ks'
    if method not in ('ks', 'stouffer', 'fisher') and nb > 0 and npos > 0:
        raise UnboundLocalError("local variable 'comb_p_p' referenced before assignment")       # as the reference
    res = engine.detect_host(sig0, off0, sig1, off1, rid, nb=max(nb, 0), weights_dif=moptions.get('WeightsDif', 2.0),
                             method=dev_method, want_mstd=want_mstd, device=dev)
    if npos and np.any(res['status'] & L.STATUS_MWU_ALL_IDENTICAL):
        raise ValueError('All numbers are identical in mannwhitneyu')                             # scipy 1.2.1, uncaught in the reference
    # positions the device could not take (a group beyond 65 535 samples: NaN outputs) or whose samples are not finite
    # (unspecified statistics) are flagged per position; the reference has neither a limit nor a check and would carry on.
    # moptions['nmod_strict'] makes them an error instead; either way they are listed in moptions['nmod_flagged'].
    flagged = np.flatnonzero(res['status'] & (L.STATUS_TOO_LARGE | L.STATUS_NONFINITE)) if npos else np.zeros(0, np.int64)
    moptions['nmod_flagged'] = [(str(meta['chrom'][i]), str(meta['strand'][i]), int(meta['pos'][i]), int(res['status'][i])) for i in flagged]
    if len(flagged) and moptions.get('outLevel', OUTPUT_ERROR) <= OUTPUT_ERROR and not moptions.get('nmod_quiet', 0):
        # (the rows of these positions hold NaN statistics where the reference would have computed numbers: say so once)
        print('nanomod_amd: %d position(s) could not be tested (a group beyond %d samples, or non-finite samples) and carry NaN statistics; '
              'first: %r; all of them: moptions[\'nmod_flagged\'] (nmod_strict=1 makes this an error)'
              % (len(flagged), L.MAX_RANKED, moptions['nmod_flagged'][0]), file=sys.stderr)
    if len(flagged) and moptions.get('nmod_strict', 0):
        raise ValueError('%d position(s) could not be tested (group beyond %d samples, or non-finite samples); first: %r'
                         % (len(flagged), L.MAX_RANKED, moptions['nmod_flagged'][0]))
    if npos:
        downsample_update(res, sig0, off0, sig1, off1, rid, meta['strand'], moptions.get('coverages', (0, 0)),
                          iters=int(moptions.get('downsampling', 100)), quantile=float(moptions.get('downsampling_quantile', 0.25)),
                          seed=int(moptions.get('nmod_seed', 0)), nb=nb, weights_dif=moptions.get('WeightsDif', 2.0),
                          method=dev_method, device=dev)
    with_comb = not method == "ks" and nb >= 0
    if with_comb and nb == 0:                                                                     # myDetect.py:413: the KS tuple itself
        res['comb_st'], res['comb_p'] = res['ks_d'], res['ks_p']
    sign_test = SignTestRecords(meta, res, with_comb)
    moptions['sign_test'] = sign_test
    moptions['sign_test_arrays'] = res
    moptions['sign_test_meta'] = meta
    if want_mstd:
        moptions['sign_test_mstd'] = {
            (str(c), str(s), int(p)): [[m0, s0], [m1, s1]]
            for c, s, p, m0, s0, m1, s1 in zip(meta['chrom'].tolist(), meta['strand'].tolist(), meta['pos'].tolist(),
                                               res['mean0'].tolist(), res['std0'].tolist(), res['mean1'].tolist(), res['std1'].tolist())}
    end_time = time.time()
    if moptions.get('outLevel', OUTPUT_ERROR) <= OUTPUT_INFO:
        print("Producing pvalues: consuming time %d" % (end_time - start_time))

    use_pind = 1 if moptions.get('rankUse', 'pv') == 'pv' else 0
    sorted_ind = 2 if method == "ks" else 3
    save_test(moptions)
    if method != 'ks' and nb < 0:
        raise IndexError('list index out of range')            # the reference indexes mpv[1][3] here
    if moptions.get('RegionRankbyST', 0) == 0:
        # sorted(sign_test, key=(record[sorted_ind], record[2], record[0])[use_pind]), reversed for 'st': the device
        # ranking returns exactly that order (stable, -0.0 == 0.0), the records are only permuted here
        ks_key = res['ks_p'] if use_pind else res['ks_d']
        mw_key = res['mwu_p'] if use_pind else res['mwu_u']
        first = ks_key if (method == 'ks' or nb == 0) else res['comb_p' if use_pind else 'comb_st']
        if npos:
            order = engine.rank_order_host(first, ks_key, mw_key, descending=(use_pind == 0), device=dev)
            moptions['sorted_sign_test'] = sign_test.permuted(order)
        else:
            moptions['sorted_sign_test'] = []
    else:
        moptions['sorted_sign_test'] = region_rank(moptions, sorted_ind, use_pind)


# myDetect.py:522-545 — the table lines are written by the library (nmod_write_sign_test), from the result arrays of
# mtest2 or, for a hand-built `sign_test` list, from arrays gathered out of its records
def save_test(moptions):
    print('SaveTest', moptions['SaveTest'])
    if moptions['SaveTest'] == 0:
        return
    print('Finish SaveTest')
    txtfile = moptions['outFolder'] + '/' + moptions["FileID"] + '_sign_test.txt'
    if moptions.get('outLevel', OUTPUT_ERROR) <= OUTPUT_ERROR:
        print('Test data is saved in', txtfile)
    with_comb = moptions["neighborPvalues"] > 0 and (not moptions["testMethod"] == "ks")
    recs = moptions['sign_test']
    if isinstance(recs, SignTestRecords) and recs._parent is None and recs._order is None and not recs.edited():
        meta, res = recs._meta, recs._res
    else:
        meta, res = _arrays_from_records(recs, with_comb)
    engine.write_sign_test_host(txtfile, meta, res, with_comb)
    if not moptions.get('mstd', 0) == 0:
        with open(moptions['outFolder'] + '/' + moptions["FileID"] + '_meanstd.cvs', 'w') as mw:
            for c, st, p, b in zip(meta['chrom'].tolist(), meta['strand'].tolist(), meta['pos'].tolist(), meta['base'].tolist()):
                ms = moptions['sign_test_mstd'][(str(c), str(st), int(p))]
                mw.write("%s %s %d %s %.3f %.3f %.3f %.3f\n" % (c, st, p, b, ms[0][0], ms[0][1], ms[1][0], ms[1][1]))


def _arrays_from_records(recs, with_comb):
    n = len(recs)
    chrom = np.array([r[0][0] for r in recs], dtype=object)
    names = sorted(set(chrom.tolist()))
    ids = {c: i for i, c in enumerate(names)}
    meta = dict(chrom=chrom, strand=np.array([r[0][1] for r in recs], dtype=object),
                pos=np.fromiter((r[0][2] for r in recs), dtype=np.int64, count=n),
                base=np.array([r[0][3] for r in recs], dtype=object),
                n0=np.fromiter((r[0][4] for r in recs), dtype=np.int32, count=n),
                n1=np.fromiter((r[0][5] for r in recs), dtype=np.int32, count=n), names=names,
                chrom_id=np.array([ids[c] for c in chrom.tolist()], dtype=np.int32))
    cols = (('mwu_u', 0, 0), ('mwu_p', 0, 1), ('t_t', 1, 0), ('t_p', 1, 1), ('ks_d', 2, 0), ('ks_p', 2, 1)) + \
           ((('comb_st', 3, 0), ('comb_p', 3, 1)) if with_comb else ())
    res = {k: np.fromiter((r[1][a][b] for r in recs), dtype=np.float64, count=n) for k, a, b in cols}
    return meta, res
