#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W [--config ecoli|alltests|chr20|ragged]

One "step" = one pass of the hot path over this job's synthetic genome: K1 rank statistics + K2
p-values + K3 window combine on every rank's positions and — for N > 1 — the RCCL all-gather that
reassembles the per-base KS-p and combined-p tracks on every rank (BASELINE.json north_star).
Inputs are resident in HBM when the timed region starts.

Workloads (`--config`, BASELINE.json configs[1..4]; per GPU, weak scaling by default):
  ecoli     configs[1]  4.6 M positions x 200 v 200 reads, KS + weighted Stouffer window 5    (the headline, default)
  alltests  configs[2]  the same input, KS + Mann-Whitney + Welch-t + Fisher
  chr20     configs[3]  8 M positions x 500 v 500 reads per GPU (64 M over 8 GPUs), KS + Stouffer
  ragged    configs[4]  10 M positions, n0 ~ LogNormal(ln 1000, 0.5) in [5, 4000], n1 ~ LogNormal(ln 50, 0.5) in [5, 400], CSR
`--all-tests` switches any of them to the three tests + Fisher; `--dtype i16` to int16 milli-unit samples;
`--ties real` to signals on the 3-decimal grid of real NanoMod events (≈11 ties between the groups per position).

N > 1: `python3 bench.py --gpus N` starts its own N ranks (one fresh child process per GPU through
torch.distributed.run, before this process has touched a GPU); under an external launcher (WORLD_SIZE set) it is a
rank.  Positions are dealt block-cyclic (nanomod_amd/sharding.py: `chunks` rounds of N equal blocks, each block with
a +-nb halo of recomputed neighbours), so the all-gather of round c lands as one contiguous piece of the full track
and runs on RCCL's stream beside the kernels of round c+1.  Two figures are timed: `value` = kernels + all-gather
(for N=1 there is nothing to gather), and `compute_only` = the kernels alone.

Before the warm-up one untimed pass is checked against the CPU oracle (the C restatement) on a bounded sample of the
same input; the result goes into `verify`, and a failed verification makes the exit code non-zero.  The oracle is
also the `cpu_baseline` (C port on all usable cores and on one; the reference-shaped Python loop on all cores, one spawn
child per core started before this process touches a GPU).  `value` is the library default (D bit for bit).  The default
N = 1 run also times — each checked against the oracle first — all three tests + Fisher and int16 rows on the same buffers,
the rational-D flag, 3-decimal input, one GPU's share of chr20, the ragged preset with KS and with all tests, and the
host-resident entry (NMOD_MEM_HOST) on the same rows against the pinned-copy rate measured in the same run (`host_path`).
Prints ONE JSON line (the last line of stdout) on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NB = 2
WDIF = 2.0
SEED = 20240601
PLANT_PERIOD = 10000
PLANT_SHIFT = 0.8
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GUIDE_GBS = 6290.0    # same guide: measured float4 copy
# profiles/HISTORY.md B.3: instruction budget of the design per position (VALU wave-instructions), by (mode, n0, n1)
FLOOR_INSTR = {('ks', 200, 200): 342 + 39, ('all', 200, 200): 640}
KS_D_RATIONAL_ABS = 4.5e-16    # gate on D under NMOD_FLAG_KS_RATIONAL_D (2 ulp); without the flag D is bit for bit

PRESETS = {
    'ecoli': dict(positions=4_600_000, n0=200, n1=200, all_tests=False, layout='stride',
                  name='BASELINE.json configs[1]: E. coli 4.6 Mb'),
    'alltests': dict(positions=4_600_000, n0=200, n1=200, all_tests=True, layout='stride',
                     name='BASELINE.json configs[2]: E. coli 4.6 Mb, all three tests'),
    'chr20': dict(positions=8_000_000, n0=500, n1=500, all_tests=False, layout='stride',
                  name='BASELINE.json configs[3]: human chr20 64 Mb over 8 GPUs = 8 M positions per GPU'),
    'ragged': dict(positions=10_000_000, n0=1000, n1=50, all_tests=False, layout='csr',
                   name='BASELINE.json configs[4]: skewed ragged coverage, 10 Mb'),
}
RAGGED_CLIP0, RAGGED_CLIP1 = (5, 4000), (5, 400)


def algorithmic_bytes(n0, n1, sample_bytes, k_out, csr):
    """s*(n0+n1) [samples] + 16 [CSR offsets; 0 for the fixed-stride layout] + 4 [run id]
    + 16*k_out [(stat, p) fp64 pairs written] + 8 [own-p re-read by the combine]."""
    return sample_bytes * (n0 + n1) + (16 if csr else 0) + 4 + 16 * k_out + 8


def usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                         # cgroup v2 CPU quota, if any
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ragged_sizes(seed, pos_begin, npos, group):
    """Group sizes of configs[4] as a function of the GLOBAL position (every rank derives the same sizes for a halo
    position): counter-based uniforms -> Box-Muller -> round(LogNormal(ln mu, 0.5)) clipped."""
    import numpy as np
    mu, (lo, hi) = ((1000.0, RAGGED_CLIP0), (50.0, RAGGED_CLIP1))[group]
    with np.errstate(over='ignore'):
        x = (np.arange(npos, dtype=np.uint64) + np.uint64(pos_begin)) * np.uint64(2) + np.uint64(group)
        x = x * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u1 = ((x >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24)
    u2 = ((x & np.uint64(0xffffff)).astype(np.float64) + 0.5) / float(1 << 24)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return np.clip(np.rint(mu * np.exp(0.5 * z)), lo, hi).astype(np.int64)


def synth_rows(seed, pos_begin, npos, group, n_per_pos, plant_period, plant_shift, i16):
    """numpy restatement of the device generator (include/nanomod_hip.h: nmod_synth_fill / nmod_synth_fill_csr; bit-equal,
    tests/test_gpu_parity.py) as an [npos, n_per_pos] array: the CPU legs make the workload's rows without a GPU."""
    import numpy as np
    with np.errstate(over='ignore'):
        pos = (np.arange(npos, dtype=np.int64) + pos_begin)[:, None]
        read = np.arange(n_per_pos, dtype=np.uint64)[None, :]
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (pos * 2 + group).astype(np.uint64)
        x = x ^ (read * np.uint64(0xD1B54A32D192ED03))
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    m = np.uint64(0xffff)
    sm = ((x & m) + ((x >> np.uint64(16)) & m) + ((x >> np.uint64(32)) & m) + (x >> np.uint64(48))).astype(np.int64)
    v = (sm - 131070).astype(np.float32) * np.float32(2.6428997e-05)
    if group == 1 and plant_period > 0:
        mm = pos % plant_period
        v = np.where((mm == 0) | (mm == 1) | (mm == plant_period - 1), v + np.float32(plant_shift), v).astype(np.float32)
    if i16:
        return np.rint(v * np.float32(1000.0)).astype(np.int16)
    return v


def synth_event_rows(seed, pos_begin, npos, group, n_per_pos, plant_period, plant_shift_milli, spread_milli, i16, outlier_permille=0):
    """numpy restatement of nmod_synth_fill_events (include/nanomod_hip.h; bit-equal, tests/test_gpu_parity.py) as an
    [npos, n_per_pos] array: a level per position shared by both groups, reads spread around it, on the milli-unit grid."""
    import numpy as np

    def mix(sd, pos, group, read):
        with np.errstate(over='ignore'):
            x = np.uint64(sd) + np.uint64(0x9E3779B97F4A7C15) * (pos * 2 + group).astype(np.uint64)
            x = x ^ (read * np.uint64(0xD1B54A32D192ED03))
            x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
            x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
        return x
    pos = (np.arange(npos, dtype=np.int64) + pos_begin)[:, None]
    read = np.arange(n_per_pos, dtype=np.uint64)[None, :]
    lev = ((mix(np.uint64(seed) ^ np.uint64(0xA5A5A5A5DEADBEEF), pos, 0, np.zeros((1, 1), np.uint64)) >> np.uint64(40)) % np.uint64(6001)).astype(np.int64) - 3000
    x = mix(seed, pos, group, read)
    m = np.uint64(0xffff)
    z = ((x & m) + ((x >> np.uint64(16)) & m) + ((x >> np.uint64(32)) & m) + (x >> np.uint64(48))).astype(np.int64) - 131070
    k = lev + np.floor_divide(2 * z * int(spread_milli) + 37837, 75674)
    if group == 1 and plant_period > 0:
        mm = pos % plant_period
        k = k + np.where((mm == 0) | (mm == 1) | (mm == plant_period - 1), int(plant_shift_milli), 0)
    if outlier_permille > 0:
        o = mix(np.uint64(seed) ^ np.uint64(0x0DDBA11C0FFEE123), pos, group, read)
        hit = ((o >> np.uint64(20)) % np.uint64(1000)).astype(np.int64) < int(outlier_permille)
        k = np.where(hit, ((o >> np.uint64(32)) % np.uint64(10001)).astype(np.int64) - 5000, k)
    k = np.clip(k, -32767, 32767)
    return k.astype(np.int16) if i16 else (k.astype(np.float64) / 1000.0).astype(np.float32)


def refpy_worker(idx, go, results, cfg):
    """One process of the reference-shaped CPU leg: the reference's own shape of the computation — per position,
    in Python, the three scipy-1.2.1-style tests of getKStest (myDetect.py:327-343) through oracle/nanomod_oracle.py, then
    the window combine over the process's track (myDetect.py:373-414).  Started before the parent touches a GPU; never
    touches one itself; idles on `go` until the parent's CPU-baseline leg."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import nanomod_oracle as orc
    orc.getKStest(np.arange(8.0), np.arange(8.0) + 0.5)                       # imports and first-call work done before the clock
    go.wait()
    t0 = time.time()
    seed, n0, n1, csr, i16, method = cfg['seed'], cfg['n0'], cfg['n1'], cfg['csr'], cfg['i16'], cfg['method']
    scale = 1e-3 if i16 else 1.0
    block = 200 if csr else 1000
    pos = cfg['pos_begin'] + idx * cfg['stride']
    done, ksd, ksp = 0, [], []
    while done < cfg['min_positions'] or time.time() - t0 < cfg['budget_s']:
        if csr:
            s0 = ragged_sizes(seed, pos, block, 0); s1 = ragged_sizes(seed, pos, block, 1)
        else:
            s0 = np.full(block, n0); s1 = np.full(block, n1)
        if cfg.get('spread', 0) > 0:
            a = synth_event_rows(seed, pos, block, 0, int(s0.max()), PLANT_PERIOD, int(round(PLANT_SHIFT * 1000)), cfg['spread'], i16, cfg.get('outliers', 0))
            b = synth_event_rows(seed, pos, block, 1, int(s1.max()), PLANT_PERIOD, int(round(PLANT_SHIFT * 1000)), cfg['spread'], i16, cfg.get('outliers', 0))
        else:
            a = synth_rows(seed, pos, block, 0, int(s0.max()), PLANT_PERIOD, PLANT_SHIFT, i16)
            b = synth_rows(seed, pos, block, 1, int(s1.max()), PLANT_PERIOD, PLANT_SHIFT, i16)
        for i in range(block):
            r = orc.getKStest(a[i, :s0[i]].astype(np.float64) * scale, b[i, :s1[i]].astype(np.float64) * scale)
            ksd.append(r[2][0]); ksp.append(r[2][1])
        pos += block; done += block
        if done >= cfg['max_positions']:
            break
    orc.combine_track(np.array(ksd), np.array(ksp), np.zeros(done, np.int32), NB, WDIF,
                      orc.METHOD_STOUFFER if method == 'stouffer' else orc.METHOD_FISHER)
    results.put((idx, done, t0, time.time()))


def start_refpy_workers(cfg, procs):
    """spawn (fresh interpreters, nothing inherited) the reference-shaped workers; they import, warm up and wait"""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    # the children are CPU-only: they get no profiler preload (this run may sit under rocprofv3) and see no GPU.  The first
    # SemLock starts multiprocessing's resource_tracker process, so Event / Queue are made AFTER the environment is scrubbed.
    saved = dict(os.environ)
    try:
        for k in list(os.environ):
            if k == 'LD_PRELOAD' or k.startswith(('ROCP', 'ROCPROF', 'HSA_TOOLS', 'ROCTX', 'RPD_')):
                del os.environ[k]
        os.environ['ROCR_VISIBLE_DEVICES'] = ''
        os.environ['HIP_VISIBLE_DEVICES'] = ''
        go, results = ctx.Event(), ctx.Queue()
        ps = [ctx.Process(target=refpy_worker, args=(i, go, results, cfg), daemon=True) for i in range(procs)]
        for p_ in ps:
            p_.start()
    finally:
        os.environ.clear()
        os.environ.update(saved)
    return {'go': go, 'results': results, 'procs': ps, 'cfg': cfg}


def collect_refpy(h):
    import queue
    h['go'].set()
    got = []
    deadline = time.time() + 900
    while len(got) < len(h['procs']):
        try:
            got.append(h['results'].get(timeout=2))
        except queue.Empty:
            dead = [p_ for p_ in h['procs'] if not p_.is_alive() and p_.exitcode not in (0, None)]
            if dead or time.time() > deadline:
                raise RuntimeError('reference-shaped CPU leg: %d worker(s) failed' % max(len(dead), 1))
    for p_ in h['procs']:
        p_.join(timeout=30)
    positions = sum(g[1] for g in got)
    wall = max(g[3] for g in got) - min(g[2] for g in got)
    return positions, wall, min(g[1] for g in got)


def oracle_run(a, off0, b, off1, npos, method, tests, threads):
    """The checker: oracle/nanomod_oracle.c on the first `npos` rows (CSR views of a, b).  Returns (outputs, seconds)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import oracle_c
    t0 = time.perf_counter()
    out = oracle_c.detect_batch(a[:off0[npos]], off0[:npos + 1], b[:off1[npos]], off1[:npos + 1], np.zeros(npos, np.int32),
                                NB, WDIF, method, tests=tests, threads=threads)
    return out, time.perf_counter() - t0


def cpu_baseline(rows, what, method, tests, threads, target_seconds=6.0, refpy=None):
    """The oracle timed on this box's host cores on a bounded sample of the same workload (the first rows of rank 0's
    device-resident input, copied back): all usable cores, and one core."""
    import numpy as np
    a, off0, b, off1 = rows
    cap = len(off0) - 1
    oracle_run(a, off0, b, off1, min(2000, cap), method, tests, threads)                 # thread-pool warm-up
    probe = min(20000, cap)
    rate = probe / oracle_run(a, off0, b, off1, probe, method, tests, threads)[1]
    sample = int(min(cap, max(probe, rate * target_seconds)))
    reps = max(1, int(round(rate * target_seconds / sample)))
    dt = sum(oracle_run(a, off0, b, off1, sample, method, tests, threads)[1] for _ in range(reps))
    one = int(min(cap, max(2000, rate / max(threads, 1) * 2.0)))                      # ~2 s on one core
    dt1 = oracle_run(a, off0, b, off1, one, method, tests, 1)[1]
    # the reference's own shape of the computation — one scipy-style call sequence per position in Python, all three tests as
    # getKStest always computes them — on every usable core: one process per core, >= 20 000 positions each
    if refpy is not None:
        positions, wall, least = collect_refpy(refpy)
        procs = len(refpy['procs'])
        ref_shaped = {'value': positions / wall, 'unit': 'positions/s', 'cores': procs, 'positions': positions, 'wall_seconds': wall,
                      'sample': 'oracle/nanomod_oracle.py getKStest (MWU + Welch-t + KS, the scipy 1.2.1 formulas) per position + the window '
                                'combine, %d processes (multiprocessing spawn, started before the first GPU call) x >= %d positions of the same '
                                'generator (%s), %d positions in %.1f s wall; the whole workload at this rate: %.0f s'
                                % (procs, least, what, positions, wall, refpy['cfg']['workload_positions'] / (positions / wall))}
    else:
        import nanomod_oracle as orc
        npy = min(300, cap)
        scale = 1e-3 if a.dtype == np.int16 else 1.0
        t0 = time.perf_counter()
        ksp = [orc.getKStest(a[off0[i]:off0[i + 1]].astype(np.float64) * scale, b[off1[i]:off1[i + 1]].astype(np.float64) * scale)[2][1]
               for i in range(npy)]
        orc.combine_track(np.zeros(npy), np.array(ksp), np.zeros(npy, np.int32), NB, WDIF,
                          orc.METHOD_STOUFFER if method == 'stouffer' else orc.METHOD_FISHER)
        dpy = time.perf_counter() - t0
        ref_shaped = {'value': npy / dpy, 'unit': 'positions/s', 'cores': 1, 'positions': npy, 'wall_seconds': dpy,
                      'sample': 'oracle/nanomod_oracle.py getKStest + combine per position on the first %d positions, one core' % npy}
    return {'value': sample * reps / dt, 'unit': 'positions/s', 'cores': threads, 'kind': 'port', 'cpu_model': cpu_model(),
            'sample_short': 'first %d positions of the same workload x %d passes, oracle/nanomod_oracle.c (OpenMP, %d threads), %.1f s' % (sample, reps, threads, dt),
            'sample': 'first %d positions of the same workload (%s) x %d passes, oracle/nanomod_oracle.c '
                      'with OpenMP on %d threads (cgroup CPU quota), %.1f s' % (sample, what, reps, threads, dt),
            'one_core': {'value': one / dt1, 'unit': 'positions/s', 'cores': 1,
                         'sample': 'first %d positions, same library, 1 thread, %.1f s' % (one, dt1)},
            'reference_shaped_python': ref_shaped}


def drop_in_leg(nm, dev_index, positions=460_000, reads=20, shapes=('arrays', 'lists'), split=False, event_like=False):
    """The function-level drop-in on the reference's own in-memory shape (myDetect.py:569-572: dict[(chrom, strand)][pos] -> list of
    numpy.float64, built at :124) — mfilter_coverage + mtest2 (position set and order, CSR, PCIe, K1-K3, ranking, `_sign_test.txt`)
    at a tenth of E. coli, 20 v 20 reads of 3-decimal values, two strands; per-position numpy arrays as the second shape.
    The first 2 000 positions' p-values are checked against the oracle."""
    import contextlib
    import io
    import shutil
    import tempfile
    import numpy as np
    from ctypes import byref as ctypes_byref
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle_c
    rng = np.random.default_rng(SEED)
    half = positions // 2
    if event_like:                 # a level per position shared by the groups, reads spread 0.2 units around it (cf. nmod_synth_fill_events)
        lev = rng.uniform(-3, 3, (positions, 1))
        vals = {ds: np.round(lev + shift + 0.2 * rng.standard_normal((positions, reads), dtype=np.float32), 3) for ds, shift in (('A', 0.0), ('B', 0.02))}
    else:
        vals = {ds: np.round(rng.normal(shift, 1, (positions, reads)), 3) for ds, shift in (('A', 0.0), ('B', 0.1))}
    out = {'positions': positions, 'reads_per_group': reads, 'dtype': 'float64 on the 0.001 grid',
           'rows': 'event-like: level per position in +-3 units, spread 0.2' if event_like else 'N(0, 1) / N(0.1, 1)',
           'note': 'nanomod_amd.mfilter_coverage + nanomod_amd.mtest2 (testMethod stouffer, neighborPvalues 2, SaveTest 1) on the reference\'s '
                   'dict shape; seconds are wall time of the two calls, the input dicts are built outside the clock'}
    for shape in shapes:
        tmp = tempfile.mkdtemp()
        mo = {'ds2': ['A', 'B'], 'outLevel': 3, 'mstd': 0, 'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25,
              'neighborPvalues': NB, 'WeightsDif': WDIF, 'testMethod': 'stouffer', 'rankUse': 'pv', 'SaveTest': 1, 'RegionRankbyST': 0,
              'outFolder': tmp, 'FileID': 'bench', 'MinCoverage': 5, 'nmod_device': dev_index}
        for ds in ('A', 'B'):
            v = vals[ds]
            row = (lambda i: v[i]) if shape == 'arrays' else (lambda i: [np.float64(x) for x in v[i]])
            mo[ds] = {'norm_mean': {('chr', '+'): {i: row(i) for i in range(half)}, ('chr', '-'): {i: row(i) for i in range(half, positions)}},
                      'base': {('chr', '+'): {i: 'A' for i in range(half)}, ('chr', '-'): {i: 'C' for i in range(half, positions)}}, 'basedict': {}}
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            nm.mfilter_coverage(mo)
            t1 = time.perf_counter()
            nm.mtest2(mo)
            t2 = time.perf_counter()
        res = mo['sign_test_arrays']
        vn = 2000
        a = vals['A'][:vn].reshape(-1); b = vals['B'][:vn].reshape(-1)
        off = np.arange(0, (vn + 1) * reads, reads, dtype=np.int64)
        exp = oracle_c.detect_batch(np.rint(a * 1000).astype(np.int16), off, np.rint(b * 1000).astype(np.int16), off, np.zeros(vn, np.int32),
                                    NB, WDIF, 'stouffer', tests=7)
        inner = slice(0, vn - NB)
        ok = all(bool(np.all(np.abs(res[k][:vn][inner] - exp[k][inner]) <= 1e-9 * np.abs(exp[k][inner]) + 1e-300)) for k in ('mwu_p', 't_p', 'ks_p', 'comb_p'))
        ok = ok and bool(np.array_equal(res['mwu_u'][:vn], exp['mwu_u'])) and bool(np.array_equal(res['ks_d'][:vn], exp['ks_d']))
        lines = sum(1 for _ in open(os.path.join(tmp, 'bench_sign_test.txt')))
        shutil.rmtree(tmp, ignore_errors=True)
        out[shape] = {'mfilter_coverage_s': t1 - t0, 'mtest2_s': t2 - t1, 'positions_per_s': positions / (t2 - t0),
                      'table_lines': lines, 'first_ranked': list(mo['sorted_sign_test'][0][0][:3]), 'verify_ok': bool(ok and lines == positions)}
        if split:
            # the stages of mtest2 one by one, through the functions it calls (myDetect.py:416-462: walk + order, tests, ranking, table)
            from nanomod_amd import detect as D, engine as E
            tmp2 = tempfile.mkdtemp()
            st = {}
            with contextlib.redirect_stdout(io.StringIO()):
                t = time.perf_counter(); meta, s0, o0, s1, o1, rid_ = D.build_csr(mo); st['build_csr_s'] = time.perf_counter() - t
                t = time.perf_counter()
                r_ = E.detect_host(s0, o0, s1, o1, rid_, nb=NB, weights_dif=WDIF, method='stouffer', device=dev_index)
                st['detect_host_s'] = time.perf_counter() - t
                hs = nm._lib.NmodHostStats(); nm._lib.load().nmod_last_host_stats(ctypes_byref(hs))
                t = time.perf_counter(); E.rank_order_host(r_['comb_p'], r_['ks_p'], r_['mwu_p'], device=dev_index); st['rank_order_s'] = time.perf_counter() - t
                t = time.perf_counter(); E.write_sign_test_host(os.path.join(tmp2, 'x.txt'), meta, r_, True); st['write_table_s'] = time.perf_counter() - t
            st.update({'csr_dtype': str(s0.dtype), 'h2d_bytes': int(hs.h2d_bytes), 'h2d_GBps_over_detect_host': hs.h2d_bytes / st['detect_host_s'] / 1e9,
                       'chunks': int(hs.chunks)})
            out[shape]['stages'] = st
            shutil.rmtree(tmp2, ignore_errors=True)
            del r_, s0, s1
        del mo
    return out


def gpu_local_cpus(torch, dev_index):
    """CPUs of the NUMA node the GPU hangs off (sysfs), or None.  A two-socket host copies from the far socket's memory at
    ~0.8 of the near rate; where the caller's arrays live is the caller's business, the measurement keeps its own near."""
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return cpus or None
    except Exception:
        return None


def host_path_leg(nm, torch, dev_index, blocks, n0, n1, nb, wdif, method, tests, want_i16, ref_out, flags=0):
    """The host-resident entry (NMOD_MEM_HOST; what a drop-in mtest2 hands over — everything on this path is host memory in
    the reference, myDetect.py:416-445) on the SAME rows as the headline: pageable numpy arrays, the same arrays page-locked
    in place, and int16 milli-unit arrays.  The path is PCIe-bound, so its roofline is the pinned hipMemcpy rate, measured
    here in the same run.  Results are compared bit for bit with the device-resident pass."""
    import ctypes
    import numpy as np
    L = nm._lib
    lib = L.load()
    dev = 'cuda:%d' % dev_index
    b0 = blocks[0]
    npos = b0['n']
    # the host arrays of this leg are first touched by this thread: keep it (and the library's copy threads, which inherit the
    # mask) on the GPU's own socket while the leg runs
    near = gpu_local_cpus(torch, dev_index)
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        if near and (near & old_aff):
            os.sched_setaffinity(0, near & old_aff)
        else:
            near = None
    except (AttributeError, OSError):
        near = None
    try:
        return _host_path_leg(nm, torch, dev_index, blocks, n0, n1, nb, wdif, method, tests, want_i16, ref_out, near, flags)
    finally:
        if old_aff is not None:
            try:
                os.sched_setaffinity(0, old_aff)
            except OSError:
                pass


def _host_path_leg(nm, torch, dev_index, blocks, n0, n1, nb, wdif, method, tests, want_i16, ref_out, near, flags):
    import ctypes
    import numpy as np
    L = nm._lib
    lib = L.load()
    dev = 'cuda:%d' % dev_index
    b0 = blocks[0]
    npos = b0['n']
    # ---- the PCIe roofline: pinned host -> device copies of 1 GiB, HIP events on the copy stream
    pin = torch.empty(1 << 28, dtype=torch.float32).pin_memory()
    dst = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dst.copy_(pin, non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    h2d_gbs = 3 * pin.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    e0.record()
    for _ in range(3):
        pin.copy_(dst, non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    d2h_gbs = 3 * pin.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del pin, dst
    rid = np.zeros(npos, np.int32)

    def run(a, b, reps):
        best, st, res = None, None, None
        for _ in range(reps):                        # (the result arrays of the first call are written again by the next ones)
            t0 = time.perf_counter()
            res = nm.detect_host(a, None, b, None, rid, nb=nb, weights_dif=wdif, method=method, tests=tests, stride0=n0, stride1=n1, device=dev_index, out=res,
                                 flags=flags)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best = dt
                st = L.NmodHostStats()
                lib.nmod_last_host_stats(ctypes.byref(st))
        return best, st, res

    def record(label, a, b, reps, check):
        dt, st, res = run(a, b, reps)
        rec = {'input': label, 'positions_per_s': npos / dt, 'seconds': dt, 'h2d_GBps': st.h2d_bytes / dt / 1e9,
               'frac_of_pinned_h2d': st.h2d_bytes / dt / 1e9 / h2d_gbs, 'chunks': st.chunks, 'slots': st.slots,
               'copy_threads': st.copy_threads, 'pinned_input': bool(st.pinned_input), 'device_bytes': st.device_bytes,
               'h2d_bytes': st.h2d_bytes, 'd2h_bytes': st.d2h_bytes}
        if check is not None:
            rec['equals_device_resident_pass'] = bool(all(np.array_equal(res[k], check[k].cpu().numpy(), equal_nan=True) for k in res if k in check))
        return rec, res

    out = {'pinned_h2d_GBps': h2d_gbs, 'pinned_d2h_GBps': d2h_gbs, 'positions': npos, 'host_arrays_on_gpu_socket': bool(near),
           'note': 'nmod_detect_batch(NMOD_MEM_HOST) on the headline rows held in host memory: chunks of positions through pinned bounce '
                   'slots, H2D of chunk k+1 / K1+K2 of chunk k / D2H of chunk k-1 on three streams, one K3 over the whole KS track; '
                   'best of the runs listed; roofline = the pinned H2D rate measured above'}
    a = b0['sig0'].cpu().numpy(); b = b0['sig1'].cpu().numpy()
    nm.detect_host(a[:n0 * 4096], None, b[:n1 * 4096], None, rid[:4096], nb=nb, weights_dif=wdif, method=method, tests=tests, stride0=n0, stride1=n1, device=dev_index)
    dtype_name = 'float32' if a.dtype == np.float32 else 'int16'
    out['pageable_%s' % dtype_name], _ = record('pageable numpy %s' % dtype_name, a, b, 3, ref_out)
    cudart = torch.cuda.cudart()
    ok = all(int(cudart.cudaHostRegister(x.ctypes.data, x.nbytes, 0)) == 0 for x in (a, b))
    if ok:
        out['pinned_%s' % dtype_name], _ = record('the same arrays page-locked in place (hipHostRegister)', a, b, 3, ref_out)
        for x in (a, b):
            cudart.cudaHostUnregister(x.ctypes.data)
    if want_i16 and a.dtype == np.float32:
        # the same rows as int16 milli-units (the format of real events): device pass for the check, then the host entry
        det16 = nm.DeviceDetector(dev_index, nb=nb, weights_dif=wdif, method=method, tests=tests, flags=flags)
        q0 = torch.empty(npos * n0, dtype=torch.int16, device=dev); q1 = torch.empty(npos * n1, dtype=torch.int16, device=dev)
        det16.synth_fill(q0, SEED, b0['lo_h'], npos, 0, n0, PLANT_PERIOD, PLANT_SHIFT)
        det16.synth_fill(q1, SEED, b0['lo_h'], npos, 1, n1, PLANT_PERIOD, PLANT_SHIFT)
        ref16 = det16.run(q0, q1, b0['rid'], stride0=n0, stride1=n1, npos=npos)
        torch.cuda.synchronize()
        a16 = q0.cpu().numpy(); b16 = q1.cpu().numpy()
        del q0, q1
        out['pageable_int16'], _ = record('pageable numpy int16 (milli-units)', a16, b16, 3, ref16)
        # ... and as float64 on the 0.001 grid, k / 1000.0 — what build_csr sends for the reference's lists of numpy.float64
        # (myDetect.py:124) when a batch is large: 8 B per sample over the bus, keys picked per position on the device.  A quarter
        # of the positions (3.7 GB of host memory), checked against the int16 pass
        nq = npos // 4
        a64 = a16[:nq * n0].astype(np.float64) / 1000.0; b64 = b16[:nq * n1].astype(np.float64) / 1000.0
        rid_q = rid[:nq]
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r64 = nm.detect_host(a64, None, b64, None, rid_q, nb=nb, weights_dif=wdif, method=method, tests=tests, stride0=n0, stride1=n1, device=dev_index, flags=flags)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        st = L.NmodHostStats(); lib.nmod_last_host_stats(ctypes.byref(st))
        same = bool(np.array_equal(r64['ks_d'][:nq - nb], ref16['ks_d'][:nq - nb].cpu().numpy()) and
                    np.allclose(r64['ks_p'][:nq - nb], ref16['ks_p'][:nq - nb].cpu().numpy(), rtol=1e-12, atol=0))
        out['float64_grid'] = {'input': 'pageable numpy float64, k / 1000.0 (the first quarter of the positions): narrowed to int16 by the threads that fill the bounce slots',
                               'narrowed_chunks': int(st.narrowed_chunks), 'copy_threads': int(st.copy_threads), 'positions': nq,
                               'positions_per_s': nq / best, 'seconds': best, 'h2d_GBps': st.h2d_bytes / best / 1e9,
                               'frac_of_pinned_h2d': st.h2d_bytes / best / 1e9 / h2d_gbs, 'h2d_bytes': int(st.h2d_bytes), 'chunks': int(st.chunks),
                               'equals_int16_pass': same}
    return out


def profile_record(lib_path, key):
    """Counters of this workload's K1 kernel from a rocprofv3 PMC run of THIS library binary (tools/profile_round.py
    writes profiles/traffic.json keyed by workload + library sha): HBM bytes per launch, VALU instructions per position,
    VALU issue utilisation.  None for a binary that has not been profiled."""
    try:
        sha = hashlib.sha256(open(lib_path, 'rb').read()).hexdigest()[:16]
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        ent = rec.get(key)
        if ent and ent.get('lib_sha16') == sha:
            return ent
    except Exception:
        pass
    return None


def rank_plan(args, world, rank):
    """What this rank will hold in HBM (no GPU call, no library): its blocks of the block-cyclic partition + halos, their
    sample bytes (ragged: from the position-keyed size generator), workspace, result tracks and the gather buffers."""
    sys.path.insert(0, ROOT)
    from nanomod_amd import sharding
    preset = PRESETS[args.config]
    csr = preset['layout'] == 'csr'
    all_tests = bool(args.all_tests or preset['all_tests'])
    n0, n1 = args.n0 or preset['n0'], args.n1 or preset['n1']
    positions = args.positions or preset['positions']
    sb = 4 if args.dtype == 'f32' else 2
    chunks = args.chunks or (4 if world > 1 else 1)
    total = positions if args.strong else positions * world
    B = sharding.cyclic_block_len(total, world, chunks)
    total = B * world * chunks
    own, with_halo, samples = 0, 0, 0
    for c in range(chunks):
        lo, hi = sharding.cyclic_block(total, world, rank, chunks, c)
        lo_h, hi_h = sharding.halo_bounds(lo, hi, NB, total)
        own += hi - lo; with_halo += hi_h - lo_h
        if csr:
            samples += int(ragged_sizes(SEED, lo_h, hi_h - lo_h, 0).sum() + ragged_sizes(SEED, lo_h, hi_h - lo_h, 1).sum())
        else:
            samples += (hi_h - lo_h) * (n0 + n1)
    k_tracks = 8 if all_tests else 4
    dev_bytes = (samples * sb + (16 * with_halo if csr else 0) + 4 * with_halo          # rows, offsets, run ids
                 + 92 * (B + 2 * NB) + 8192                                              # workspace (one block at a time; nmod_workspace_bytes)
                 + (8 * k_tracks + 1) * with_halo                                        # result tracks + status
                 + 2 * 8 * total)                                                        # gathered ks_p / comb_p (full length on every rank)
    return {'positions_total': total, 'positions_own': own, 'positions_with_halo': with_halo, 'block': B, 'chunks': chunks,
            'sample_bytes': samples * sb, 'device_bytes': dev_bytes, 'fits_288GB': bool(dev_bytes < 288e9 * 0.9)}


SIDE_FILE_DEFAULT = os.path.join(ROOT, 'bench_side.json')
LAST_LINE_CAP = 8000            # the driver keeps an 8 KB tail of stdout: the record must fit it whole (target < 4 KB)
DEFAULT_SIDE_LEGS = ('all_tests', 'int16', 'real_spread', 'outliers', 'presets', 'host_path', 'drop_in_200')
ALL_SIDE_LEGS = ('all_tests', 'int16', 'rational_d', 'real_ties', 'real_spread', 'real_spread_sweep', 'outliers', 'presets', 'presets_event',
                 'host_path', 'drop_in', 'drop_in_200')


def _sig(x, digits=6):
    """floats of the compact record carry 6 significant digits (the side file keeps every digit)"""
    if isinstance(x, float):
        return float('%.*g' % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _leg_values(side):
    """{leg: positions/s} of every measured side leg, nested groups flattened with a dot"""
    out, ok = {}, True
    for k, v in side.items():
        if not isinstance(v, dict):
            continue
        if 'value' in v:
            out[k] = v['value']
            ok = ok and bool(v.get('verify', {}).get('ok', True))
        else:
            for k2, v2 in v.items():
                if isinstance(v2, dict) and 'value' in v2:
                    out[k + '.' + k2] = v2['value']
                    ok = ok and bool(v2.get('verify', {}).get('ok', True))
    return out, ok


def compact_record(full, side, host_path, drop_in, side_file):
    """The LAST stdout line: the contract's keys + roofline + cpu_baseline + verify, short enough for the driver's 8 KB tail
    (tests/test_bench_launcher.py pins < 8 000 bytes on a canned full record).  Everything verbose — notes, definitions, every
    side leg with its own verification — is in `side_legs_file` and on earlier stdout lines of its own."""
    r = full['roofline']
    v = full['verify'] or {}
    errs_p = [v[k] for k in v if k.startswith('max_rel_err_') and k.endswith('_p')]
    errs_abs_p = [v[k] for k in v if k.startswith('max_abs_err_') and k.endswith('_p')]
    cb = full.get('cpu_baseline')
    rec = {k: full[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                'vs_baseline', 'dtype')}
    rec['data'] = full['data_short']
    c = full['config']
    rec['config'] = {k: c[k] for k in ('workload', 'preset', 'positions_total', 'positions_per_gpu', 'n0', 'n1', 'layout', 'ties',
                                       'spread_milli', 'outlier_permille', 'rccl_ranks') if k in c}
    rec['config']['ks_d'] = 'rational (NMOD_FLAG_KS_RATIONAL_D)' if 'RATIONAL' in c['ks_d'] else 'float form, bit for bit'
    rec['config']['parallelism'] = c['parallelism_short']
    rec['compute_only'] = full['compute_only']['value']
    if full.get('allgather_exposed_ms'):
        rec['allgather_exposed_ms_max'] = full['allgather_exposed_ms']['max']
    rec['roofline'] = {'bound': 'hbm', 'achieved': r['achieved'], 'peak': r['peak'], 'unit': r['unit'], 'frac': r['frac'],
                       'traffic': r['traffic'], 'traffic_attached_from': r['traffic_source'], 'limited_by': 'valu-issue',
                       'kernel': r['kernel'], 'kernel_avg_ms': r['kernel_avg_ms'], 'path_avg_ms': r['path_avg_ms'],
                       'launches_timed': r['launches_timed'], 'algorithmic_bytes_per_position': r['algorithmic_bytes_per_position'],
                       'positions_per_launch': r['positions_per_launch'], 'measured_copy_GBps': r['measured_copy_GBps'],
                       'profile_key': r.get('profile_key')}
    rec['valu'] = {'instr_per_position': full['valu']['instr_per_position'], 'issue_util': full['valu']['issue_util'],
                   'attached_from': full['valu']['source']}
    if full.get('form_share') is not None:
        rec['form_share'] = full['form_share']
    rec['verify'] = {'ok': bool(v.get('ok')), 'positions': v.get('positions'), 'against': 'oracle/nanomod_oracle.c, same input',
                     'max_rel_err_p': max(errs_p) if errs_p else None, 'max_abs_err_p': max(errs_abs_p) if errs_abs_p else None,
                     'max_abs_err_ks_d': v.get('max_abs_err_ks_d'), 'max_abs_err_mwu_u': v.get('max_abs_err_mwu_u')}
    for k in ('gathered_track_equals_local', 'block_boundaries_checked', 'block_boundary_positions_differing'):
        if k in v:
            rec['verify'][k] = v[k]
    if cb is not None:
        rec['cpu_baseline'] = {'value': cb['value'], 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'],
                               'cpu_model': cb.get('cpu_model'), 'sample': cb['sample_short'],
                               'one_core': cb['one_core']['value'],
                               'reference_shaped_python': {'value': cb['reference_shaped_python']['value'],
                                                           'cores': cb['reference_shaped_python']['cores']}}
    vals, side_ok = _leg_values(side)
    if vals:
        rec['side'] = vals
        rec['side_verify_ok'] = side_ok
    if host_path is not None:
        rec['host_path'] = {k: v_['positions_per_s'] for k, v_ in host_path.items() if isinstance(v_, dict) and 'positions_per_s' in v_}
        rec['host_path']['pinned_h2d_GBps'] = host_path.get('pinned_h2d_GBps')
    if drop_in is not None:
        rec['drop_in_mtest2'] = {k: v_['positions_per_s'] for k, v_ in drop_in.items() if isinstance(v_, dict) and 'positions_per_s' in v_}
        big = drop_in.get('at_200v200', {}).get('arrays')
        if big:
            rec['drop_in_mtest2']['at_200v200'] = big['positions_per_s']
    rec['side_legs_file'] = side_file
    rec['leg_seconds'] = full.get('leg_seconds')
    rec['build_info'] = full['build_info_short']
    rec['lib_sha16'] = full.get('lib_sha16')
    rec = _sig(rec)
    text = json.dumps(rec, separators=(',', ':'))
    if len(text) >= LAST_LINE_CAP:                                   # never hand the driver a line it cannot keep
        for k in ('leg_seconds', 'drop_in_mtest2', 'host_path', 'side'):
            rec.pop(k, None)
            text = json.dumps(rec, separators=(',', ':'))
            if len(text) < LAST_LINE_CAP:
                break
    return rec, text


def short_build_info(info):
    """'arch=gfx950 abi=3 hip=7.2' + the experiment switches that differ from the shipped setting in any translation unit"""
    parts = info.split(' | ')
    head = parts[0]
    shipped = {'NMOD_SKIP': '0', 'NMOD_EXP': '0', 'NMOD_SWZ_MASK': '0', 'NMOD_PK_SELECT': '0', 'NMOD_CE_BUILTIN': '0', 'NMOD_XOR4_BANKS': '0',
               'NMOD_NO_GRID': '0', 'NMOD_CNT_SKIP': '0', 'NMOD_CW_OR3': '1', 'NMOD_CNT_TAILS': '1', 'NMOD_WIDE_TAILS': '1'}
    odd = set()
    for tu in parts[1:]:
        name, _, kv = tu.partition(': ')
        for item in kv.split():
            k, _, val = item.partition('=')
            if k in shipped and val != shipped[k]:
                odd.add('%s:%s' % (name, item))
    return head + (' | experiment switches on: ' + ' '.join(sorted(odd)) if odd else ' | experiment switches: none')


def emit_line(text):
    """One record = ONE write(2): torch.distributed.run starts its workers with `python -u`, where print() sends the body
    and the newline separately and the lines of different ranks interleave on the shared pipe."""
    sys.stdout.flush()
    data = (text + '\n').encode()
    while data:
        data = data[os.write(1, data):]


def split_json_objects(ln):
    """A forwarded line may still hold several records back to back (`{...}{...}`): cut it into its top-level objects."""
    dec, out, i = json.JSONDecoder(), [], 0
    try:
        while i < len(ln):
            _, j = dec.raw_decode(ln, i)
            out.append(ln[i:j]); i = j
            while i < len(ln) and ln[i].isspace():
                i += 1
    except ValueError:
        return [ln]
    return out or [ln]


def launch_ranks(args, argv):
    """N > 1 without an external launcher: one fresh child process per GPU.  This process has not imported torch.cuda
    or loaded the HIP library; it only forwards the children's output (rank 0's JSON line last) and their failure."""
    port = free_port()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    last_json = None
    for ln in proc.stdout:
        ln = ln.strip('\n')
        if not ln:
            continue
        if ln.startswith('{') and ln.endswith('}'):
            for rec in split_json_objects(ln):
                if last_json is not None:
                    emit_line(last_json)
                last_json = rec
        else:
            emit_line(ln)
    rc = proc.wait()
    if last_json is not None:
        emit_line(last_json)
    if rc != 0:
        print('bench.py: a rank failed (torch.distributed.run exit code %d); nothing is retried' % rc, file=sys.stderr)
    return rc if rc != 0 else (0 if last_json is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', choices=sorted(PRESETS), default='ecoli', help='BASELINE.json configs[1..4]')
    ap.add_argument('--positions', type=int, default=0, help='positions per GPU (weak) or in total (--strong); default: the preset\'s')
    ap.add_argument('--n0', type=int, default=0, help='reads per position, group 1 (default: the preset\'s; ragged: the median)')
    ap.add_argument('--n1', type=int, default=0)
    ap.add_argument('--dtype', choices=('f32', 'i16'), default='f32', help='sample dtype in HBM (i16 = milli-units)')
    ap.add_argument('--all-tests', action='store_true', help='KS + MWU + Welch-t + Fisher instead of the preset\'s tests')
    ap.add_argument('--ties', choices=('few', 'real'), default='few',
                    help='real: signals on the 3-decimal grid of NanoMod events (myRefBaseSignalAnnotation.py:1108)')
    ap.add_argument('--spread', type=int, default=0, help='event-like rows (nmod_synth_fill_events): a level per position in +-3 units, reads spread '
                    'SPREAD milli-units around it, on the 3-decimal grid (most samples of a position tie); 0 = the unit-variance generator')
    ap.add_argument('--outliers', type=int, default=0, help='with --spread: this many reads per 1 000 are mis-segmented events, a uniform draw over +-5 units '
                    '(the clip range of the raw normalisation, myRefBaseSignalAnnotation.py:251-259)')
    ap.add_argument('--no-counting', action='store_true', help='NMOD_FLAG_NO_COUNTING: every position on the sorting forms (A/B; also env NMOD_NO_COUNTING=1, read here, not by the library)')
    ap.add_argument('--no-count-wide', action='store_true', help='NMOD_FLAG_NO_COUNT_WIDE (also env NMOD_NO_COUNT_WIDE=1)')
    ap.add_argument('--strong', action='store_true', help='fixed total size: --positions in total, split over the ranks')
    ap.add_argument('--chunks', type=int, default=0, help='rounds of the block-cyclic pipeline (default 4 for N>1, 1 for N=1)')
    ap.add_argument('--force-collective', action='store_true', help='N=1: initialise RCCL with one rank and issue the all-gather anyway')
    ap.add_argument('--cpu-sample', type=int, default=0, help='cap on positions for the CPU baseline / verification (0 = 1 M)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline (the verification still runs)')
    ap.add_argument('--refpy-positions', type=int, default=20000, help='reference-shaped Python CPU leg: at least this many positions per process')
    ap.add_argument('--refpy-seconds', type=float, default=4.0, help='reference-shaped Python CPU leg: at least this many seconds per process')
    ap.add_argument('--no-side', '--no-real-ties', dest='no_side', action='store_true',
                    help='skip the side measurements of the default run (all tests, int16, rational D, tie-heavy input)')
    ap.add_argument('--side-legs', default='default', help='comma list of side measurements to run, `default` (%s) or `all` (%s)'
                    % (','.join(DEFAULT_SIDE_LEGS), ','.join(ALL_SIDE_LEGS)))
    ap.add_argument('--side-file', default=SIDE_FILE_DEFAULT, help='where the side legs (each also a stdout line of its own) and the verbose form '
                    'of the record are written; the last stdout line names it')
    ap.add_argument('--no-host-path', action='store_true', help='skip the host-resident (NMOD_MEM_HOST, PCIe-bound) measurement')
    ap.add_argument('--rational-d', action='store_true', help='KS-only configurations: time NMOD_FLAG_KS_RATIONAL_D (D as the exact rational, <= 2 ulp '
                    'from ks_2samp\'s float form) instead of the library default (D bit for bit); the default run reports this rate as a side figure')
    ap.add_argument('--launch-only', action='store_true', help='ranks print RANK / WORLD_SIZE and exit before any GPU call (launcher test)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))
    if args.launch_only:
        emit_line(json.dumps({'launch_only': True, 'rank': rank, 'local_rank': local_rank, 'world_size': world, 'plan': rank_plan(args, world, rank),
                              'master': '%s:%s' % (os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'))}))
        return

    # the reference-shaped CPU leg: one fresh process per usable core, started BEFORE this process touches a GPU (they idle
    # until the CPU-baseline leg and never load the HIP library)
    refpy = None
    if world == 1 and rank == 0 and not args.no_cpu:
        pz = PRESETS[args.config]
        at = bool(args.all_tests or pz['all_tests'])
        refpy = start_refpy_workers({'seed': SEED, 'n0': args.n0 or pz['n0'], 'n1': args.n1 or pz['n1'], 'csr': pz['layout'] == 'csr',
                                     'i16': args.dtype == 'i16', 'method': 'fisher' if at else 'stouffer', 'pos_begin': 0, 'spread': args.spread, 'outliers': args.outliers,
                                     'stride': 1_000_000, 'min_positions': args.refpy_positions, 'max_positions': 50 * args.refpy_positions,
                                     'budget_s': args.refpy_seconds, 'workload_positions': args.positions or pz['positions']}, usable_cpus())

    import numpy as np
    import torch
    import nanomod_amd as nm
    from nanomod_amd import sharding
    L = nm._lib

    form_flags = (L.FLAG_NO_COUNTING if (args.no_counting or os.environ.get('NMOD_NO_COUNTING', '0') not in ('', '0')) else 0) | \
                 (L.FLAG_NO_COUNT_WIDE if (args.no_count_wide or os.environ.get('NMOD_NO_COUNT_WIDE', '0') not in ('', '0')) else 0)
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(free_port()))
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))   # "nccl" = RCCL on ROCm
    gather = dist is not None

    preset = PRESETS[args.config]
    csr = preset['layout'] == 'csr'
    all_tests = bool(args.all_tests or preset['all_tests'])
    n0, n1 = args.n0 or preset['n0'], args.n1 or preset['n1']
    positions = args.positions or preset['positions']
    tdtype = torch.float32 if args.dtype == 'f32' else torch.int16
    sample_bytes = 4 if args.dtype == 'f32' else 2
    chunks = args.chunks or (4 if world > 1 else 1)
    total = positions if args.strong else positions * world
    B = sharding.cyclic_block_len(total, world, chunks)
    total = B * world * chunks                     # the synthetic genome is padded to whole blocks
    method = 'fisher' if all_tests else 'stouffer'
    tests = L.TEST_ALL if all_tests else L.TEST_KS
    # the timed configuration is the library default: D as ks_2samp's float form bit for bit.  --rational-d times the
    # opt-out (include/nanomod_hip.h: NMOD_FLAG_KS_RATIONAL_D, KS-only mode), which the default run reports as a side figure
    rational_d = (not all_tests) and args.rational_d
    det = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method=method, tests=tests,
                            flags=(L.FLAG_KS_RATIONAL_D if rational_d else 0) | form_flags)
    d_gate = [KS_D_RATIONAL_ABS if rational_d else 0.0]

    def fill(b, ties, spread=None, keys=('sig0', 'sig1'), outliers=None):
        """(re)generate a block's samples on the device from global position counters.  ties == 'real': the int16
        milli-unit grid (float32 input: k / 1000 as float32 — equal k <=> equal value).  spread > 0: event-like rows
        (nmod_synth_fill_events: a level per position, reads spread around it, on the grid)."""
        spread = args.spread if spread is None else spread
        outliers = args.outliers if outliers is None else outliers
        for g, key in ((0, keys[0]), (1, keys[1])):
            dst = b[key]
            if spread > 0:
                det.synth_fill_events(dst, SEED, b['lo_h'], b['n'], g, n_per_pos=0 if csr else (n0, n1)[g], off=b['off%d' % g] if csr else None,
                                      plant_period=PLANT_PERIOD, plant_shift_milli=int(round(PLANT_SHIFT * 1000)), spread_milli=spread,
                                      outlier_permille=outliers)
                continue
            grid = ties == 'real' and dst.dtype == torch.float32
            tgt = torch.empty(dst.numel(), dtype=torch.int16, device=dev) if grid else dst
            if csr:
                det.synth_fill_csr(tgt, SEED, b['lo_h'], b['off%d' % g], g, PLANT_PERIOD, PLANT_SHIFT)
            else:
                det.synth_fill(tgt, SEED, b['lo_h'], b['n'], g, (n0, n1)[g], PLANT_PERIOD, PLANT_SHIFT)
            if grid:
                # k / 1000 as a 3-decimal event value is stored: the float64 quotient (a tensor divisor: torch multiplies by the
                # reciprocal of a Python scalar, which is not the correctly rounded quotient), then rounded to float32
                step = 1 << 27
                thousand = torch.full((), 1000.0, dtype=torch.float64, device=dev)
                for lo in range(0, dst.numel(), step):
                    dst[lo:lo + step] = torch.div(tgt[lo:lo + step].to(torch.float64), thousand).to(torch.float32)
                del tgt

    # this rank's blocks (+ halo)
    blocks = []
    for c in range(chunks):
        lo, hi = sharding.cyclic_block(total, world, rank, chunks, c)
        lo_h, hi_h = sharding.halo_bounds(lo, hi, NB, total)
        n = hi_h - lo_h
        b = {'lo_h': lo_h, 'hi_h': hi_h, 'n': n, 'rid': torch.zeros(n, dtype=torch.int32, device=dev),   # one contiguous run
             'out': det.alloc_outputs(n)}
        if csr:
            for g in (0, 1):
                sz = ragged_sizes(SEED, lo_h, n, g)
                off = np.zeros(n + 1, np.int64)
                np.cumsum(sz, out=off[1:])
                b['hoff%d' % g] = off
                b['off%d' % g] = torch.from_numpy(off).to(dev)
                b['sig%d' % g] = torch.empty(int(off[-1]), dtype=tdtype, device=dev)
        else:
            b['sig0'] = torch.empty(n * n0, dtype=tdtype, device=dev)
            b['sig1'] = torch.empty(n * n1, dtype=tdtype, device=dev)
        fill(b, args.ties)
        blocks.append(b)
    n_local = sum(b['n'] for b in blocks)
    samples_local = sum(b['sig0'].numel() + b['sig1'].numel() for b in blocks)
    state = sharding.PipelinedGather(total, world, chunks, ('ks_p', 'comb_p'), dev)

    def compute(c, lo_h, hi_h):
        b = blocks[c]
        assert (lo_h, hi_h) == (b['lo_h'], b['hi_h'])
        if csr:
            return det.run(b['sig0'], b['sig1'], b['rid'], off0=b['off0'], off1=b['off1'], max_n0=RAGGED_CLIP0[1],
                           max_n1=RAGGED_CLIP1[1], out=b['out'])
        return det.run(b['sig0'], b['sig1'], b['rid'], stride0=n0, stride1=n1, npos=b['n'], out=b['out'])

    def step(with_gather):
        sharding.pipelined_detect(compute, state, NB, gather=with_gather, force_collective=args.force_collective)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    exposed = {'per_rank_ms': None}

    def timed(with_gather, steps):
        barrier()
        ev_k = torch.cuda.Event(enable_timing=True); ev_g = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(with_gather)
        ev_k.record()                                # behind the last round's kernels on the compute stream ...
        state.wait()                                 # ... which now waits for the outstanding all-gathers (none without a collective)
        ev_g.record()
        barrier()
        el = time.perf_counter() - t0
        if with_gather and dist is not None:
            # what of the gather traffic was NOT hidden behind kernels: the compute stream's wait for the last round's collectives,
            # per rank (the earlier rounds' gathers run beside the next rounds' kernels)
            mine = torch.tensor([ev_k.elapsed_time(ev_g)], dtype=torch.float64, device=dev)
            allr = torch.empty(world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(allr, mine)
            exposed['per_rank_ms'] = [float(v) for v in allr.cpu()]
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def host_rows(cap):
        """the first `cap` positions of rank 0's first block as host CSR arrays"""
        b0 = blocks[0]
        if csr:
            o0, o1 = b0['hoff0'][:cap + 1], b0['hoff1'][:cap + 1]
        else:
            o0 = np.arange(0, (cap + 1) * n0, n0, dtype=np.int64)
            o1 = np.arange(0, (cap + 1) * n1, n1, dtype=np.int64)
        return b0['sig0'][:int(o0[-1])].cpu().numpy(), o0, b0['sig1'][:int(o1[-1])].cpu().numpy(), o1

    def t_abs_gate(rows, vn):
        """absolute tolerance of the Welch statistic (tests/helpers.py: t_abs_gate): any order of summation leaves a mean within
        a few ulp, so t = (mean0 - mean1) / se is defined to ~ulp(max |mean|) / se — 1e-13 for event-like rows (|mean| ~ 3, se ~ 0.01)"""
        sc = 1e-3 if rows[0].dtype == np.int16 else 1.0
        mv = []
        for sig, off in ((rows[0], rows[1]), (rows[2], rows[3])):
            o = off[:vn + 1]
            x = sig[:int(o[-1])].astype(np.float64) * sc
            n = np.diff(o).astype(np.float64)
            idx = np.minimum(o[:-1], max(len(x) - 1, 0))
            s1 = np.add.reduceat(x, idx); s2 = np.add.reduceat(x * x, idx)
            mean = s1 / n
            mv.append((np.abs(mean), np.maximum(s2 - s1 * mean, 0.0) / np.maximum(n - 1.0, 1.0) / n))
        with np.errstate(divide='ignore', invalid='ignore'):
            g = 2e-14 + 6.0 * np.spacing(np.maximum(mv[0][0], mv[1][0])) / np.sqrt(mv[0][1] + mv[1][1])
        return np.where(np.isfinite(g), g, 2e-14)

    def verify_against_oracle(rows, vn, outs=None, leg_all=None, leg_method=None, gate=None):
        """one finished pass of rank 0's first block against the oracle on its first vn positions (tests/helpers.py gates)"""
        outs = blocks[0]['out'] if outs is None else outs
        leg_all = all_tests if leg_all is None else leg_all
        leg_method = method if leg_method is None else leg_method
        gate = d_gate[0] if gate is None else gate
        exp, _ = oracle_run(rows[0], rows[1], rows[2], rows[3], vn, leg_method, 7 if leg_all else 1, usable_cpus())
        v = {'positions': vn, 'against': 'oracle/nanomod_oracle.c on the first positions of rank 0, same input'}
        names = ['ks_d', 'ks_p', 'comb_st', 'comb_p'] + (['mwu_u', 'mwu_p', 't_t', 't_p'] if leg_all else [])
        inner = slice(0, vn - NB)                   # the sample's last nb positions see neighbours the oracle run did not
        ok = True
        for k in names:
            g = outs[k][:vn].cpu().numpy()[inner]
            e = exp[k][inner]
            fin = np.isfinite(e)
            same_special = bool(np.array_equal(g[~fin], e[~fin], equal_nan=True))
            err = np.abs(g[fin] - e[fin])
            rel = float(np.max(err / np.maximum(np.abs(e[fin]), 1e-300))) if fin.any() else 0.0
            ab = float(np.max(err)) if fin.any() else 0.0
            v['max_rel_err_' + k] = rel
            v['max_abs_err_' + k] = ab
            # the gates of tests/helpers.py: p-values 1e-9 relative and 1e-6 absolute (north_star), statistics 1e-9
            # relative + 1e-12 absolute (t: 1e-11 relative + 2e-14), U exact, D bit for bit
            if k.endswith('_p'):
                good = bool(np.all(err <= 1e-9 * np.abs(e[fin]) + 1e-300)) and ab <= 1e-6
            elif k == 'ks_d':
                good = ab <= gate
            elif k == 'mwu_u':
                good = ab == 0.0
            elif k == 't_t':
                good = bool(np.all(err <= 1e-11 * np.abs(e[fin]) + t_abs_gate(rows, vn)[inner][fin]))
            else:
                good = bool(np.all(err <= 1e-9 * np.abs(e[fin]) + 1e-12))
            ok = ok and same_special and good
        v['ok'] = bool(ok)
        return v

    def form_share(st):
        """which K1 form took the positions of a pass (nmod_last_dispatch_stats), as shares of the batch"""
        n = max(st['positions'], 1)
        out = {k: round(st[k] / n, 6) for k in ('ks_rank', 'rank_hist', 'rank_hist_wide', 'rank_pair', 'rank_count', 'rank_count_wide', 'big', 'skipped') if st[k]}
        out['counting_rejected'] = round(st['count_rejected'] / n, 6)
        return out

    # ---- one untimed pass, checked on rank 0 against the CPU oracle on a bounded sample of the same input
    step(gather)
    state.wait()
    torch.cuda.synchronize()
    headline_share = form_share(det.dispatch_stats())          # (of the pass's last block)
    verify = None
    cpu_rows = None
    mean_n = samples_local / max(n_local, 1)
    if rank == 0:
        cap_default = 1_000_000 if not csr else 200_000
        cap = min(args.cpu_sample or cap_default, blocks[0]['n'])
        cpu_rows = host_rows(cap)
        vn = min(cap, 200_000 if not csr else 40_000)
        verify = verify_against_oracle(cpu_rows, vn)
        if gather:                                   # the gathered track holds rank 0's first block at its natural place
            full = state.result()
            same = bool(torch.equal(full['ks_p'][blocks[0]['lo_h']:blocks[0]['lo_h'] + vn], blocks[0]['out']['ks_p'][:vn]))
            verify['gathered_track_equals_local'] = same
            verify['ok'] = verify['ok'] and same
            # ... and every OTHER rank's blocks at theirs: rank 0 regenerates (the generator is keyed by the global position) a
            # window of +-2 nb positions around every block boundary of the cyclic partition, recomputes it with +-nb more as
            # context, and compares with the gathered tracks bit for bit — a block landing at the wrong offset cannot pass
            W, ctx = 4 * NB, NB
            bounds = [k * B for k in range(1, world * chunks) if k * B < total]
            if bounds:
                wins = [(max(pb - W // 2 - ctx, 0), min(pb + W // 2 + ctx, total)) for pb in bounds]
                npw = sum(hi - lo for lo, hi in wins)
                w_rid = torch.cat([torch.full((hi - lo,), i, dtype=torch.int32) for i, (lo, hi) in enumerate(wins)]).to(dev)
                if csr:
                    sz = [np.concatenate([ragged_sizes(SEED, lo, hi - lo, g) for lo, hi in wins]) for g in (0, 1)]
                    w_off = [np.zeros(npw + 1, np.int64), np.zeros(npw + 1, np.int64)]
                    for g in (0, 1):
                        np.cumsum(sz[g], out=w_off[g][1:])
                    w_sig = [torch.empty(int(w_off[g][-1]), dtype=tdtype, device=dev) for g in (0, 1)]
                    at = 0
                    for lo, hi in wins:
                        for g in (0, 1):
                            o = torch.from_numpy(w_off[g][at:at + hi - lo + 1] - w_off[g][at]).to(dev)
                            det.synth_fill_csr(w_sig[g][int(w_off[g][at]):int(w_off[g][at + hi - lo])], SEED, lo, o, g, PLANT_PERIOD, PLANT_SHIFT)
                        at += hi - lo
                    wr = det.run(w_sig[0], w_sig[1], w_rid, off0=torch.from_numpy(w_off[0]).to(dev), off1=torch.from_numpy(w_off[1]).to(dev),
                                 max_n0=RAGGED_CLIP0[1], max_n1=RAGGED_CLIP1[1])
                else:
                    w_sig = [torch.empty(npw * (n0, n1)[g], dtype=tdtype, device=dev) for g in (0, 1)]
                    at = 0
                    for lo, hi in wins:
                        for g in (0, 1):
                            nn = (n0, n1)[g]
                            det.synth_fill(w_sig[g][at * nn:(at + hi - lo) * nn], SEED, lo, hi - lo, g, nn, PLANT_PERIOD, PLANT_SHIFT)
                        at += hi - lo
                    wr = det.run(w_sig[0], w_sig[1], w_rid, stride0=n0, stride1=n1, npos=npw)
                torch.cuda.synchronize()
                bad, at = 0, 0
                for (lo, hi), pb in zip(wins, bounds):
                    a0, a1 = max(pb - W // 2, 0), min(pb + W // 2, total)         # positions whose +-nb window the recomputation holds
                    for k in ('ks_p', 'comb_p'):
                        g_ = full[k][a0:a1]; e_ = wr[k][at + a0 - lo: at + a1 - lo]
                        bad += int((~((g_ == e_) | (torch.isnan(g_) & torch.isnan(e_)))).sum().item())
                    at += hi - lo
                verify['block_boundaries_checked'] = len(bounds)
                verify['block_boundary_positions_differing'] = bad
                verify['ok'] = verify['ok'] and bad == 0
        if not verify['ok']:
            print('bench.py: verification against the oracle FAILED: %r' % verify, file=sys.stderr)

    # ---- W warm-up steps, then exactly K timed steps (kernels + all-gather when N > 1)
    for _ in range(args.warmup):
        step(gather)
    timer = nm.EventTimer(max(args.steps, 1) * chunks + 8)
    det.timer = timer
    elapsed = timed(gather, args.steps)
    det.timer = None
    compute_elapsed = elapsed
    if gather:                                       # second figure: the kernels alone, same K
        compute_elapsed = timed(False, args.steps)

    k1_ms, k1_n = timer.read(L.KERNEL_RANK_STATS)
    k2_ms, _ = timer.read(L.KERNEL_FINALIZE)
    k3_ms, _ = timer.read(L.KERNEL_COMBINE)

    # measured device copy bandwidth (read + write bytes over the time of dst.copy_(src), 1 GiB each way)
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        copy_gbs = 2 * src.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    # ---- side measurements of the default run (N = 1): the same step, up to 10 timed steps each, every one checked against the
    # oracle on 20 000 positions before it is timed; `form_share`: which K1 form took the positions (nmod_last_dispatch_stats)
    def side_leg(det_x, sig_keys, outs, leg_all, leg_method, gate, note):
        def run_once():
            for b in blocks:
                det_x.run(b[sig_keys[0]], b[sig_keys[1]], b['rid'], stride0=n0, stride1=n1, npos=b['n'], out=outs)
        run_once(); torch.cuda.synchronize()
        share = form_share(det_x.dispatch_stats())
        cap = min(20_000, blocks[0]['n'])
        o0 = np.arange(0, (cap + 1) * n0, n0, dtype=np.int64); o1 = np.arange(0, (cap + 1) * n1, n1, dtype=np.int64)
        rows = (blocks[0][sig_keys[0]][:cap * n0].cpu().numpy(), o0, blocks[0][sig_keys[1]][:cap * n1].cpu().numpy(), o1)
        v = verify_against_oracle(rows, cap, outs, leg_all, leg_method, gate)
        for _ in range(2):
            run_once()
        tm = nm.EventTimer(64)
        det_x.timer = tm
        ks = max(1, min(args.steps, 10))
        barrier()
        t0 = time.perf_counter()
        for _ in range(ks):
            run_once()
        barrier()
        el = time.perf_counter() - t0
        det_x.timer = None
        k1, kn = tm.read(L.KERNEL_RANK_STATS); k2, _ = tm.read(L.KERNEL_FINALIZE); k3, _ = tm.read(L.KERNEL_COMBINE)
        sb = 2 if blocks[0][sig_keys[0]].dtype == torch.int16 else 4
        algo_leg = algorithmic_bytes(n0, n1, sb, 4 if leg_all else 2, False)
        gbs = algo_leg * n_local / ((k1 + k2 + k3) / max(kn, 1) * 1e-3) / 1e9
        if not v['ok']:
            verify['ok'] = False
            print('bench.py: verification of a side measurement FAILED (%s): %r' % (note, v), file=sys.stderr)
        return {'value': total * ks / el, 'unit': 'positions/s', 'steps': ks, 'ms_per_step': el / ks * 1e3,
                'kernel_avg_ms': k1 / max(kn, 1), 'algorithmic_bytes_per_position': algo_leg,
                'roofline_frac': gbs / HBM_PEAK_GBS, 'achieved_GBps': gbs, 'form_share': share, 'note': note, 'verify': v}

    leg_seconds = {}

    class SideLegs(dict):
        """a finished side leg is a stdout line of its own at once (a crash later in the run does not lose it) and an entry of the side file"""
        def __setitem__(self, k, v):
            dict.__setitem__(self, k, v)
            if rank == 0:
                emit_line(json.dumps({'side_leg': k, 'record': v}))

    side = SideLegs()
    legs = set(ALL_SIDE_LEGS) if args.side_legs == 'all' else set(DEFAULT_SIDE_LEGS) if args.side_legs == 'default' else set(args.side_legs.split(','))
    if 'real_spread_sweep' in legs:
        legs.add('real_spread')
    if 'presets_event' in legs:
        legs.add('presets')
    t_leg = [time.perf_counter()]

    def leg_done(name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        leg_seconds[name] = round(now - t_leg[0], 2)
        t_leg[0] = now

    leg_done('headline')
    simple = world == 1 and not csr and not args.force_collective and not args.no_side and chunks == 1
    headline_default = simple and args.config == 'ecoli' and args.dtype == 'f32' and args.ties == 'few' and not all_tests and args.spread == 0
    if headline_default and 'all_tests' in legs:
        # (a) all three tests + Fisher on the same buffers: what every real getKStest call computes (myDetect.py:331-343), BASELINE configs[2]
        det_all = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method='fisher', tests=L.TEST_ALL, flags=form_flags)
        outs_all = det_all.alloc_outputs(blocks[0]['n'])
        side['all_tests'] = side_leg(det_all, ('sig0', 'sig1'), outs_all, True, 'fisher', 0.0,
                                     'BASELINE configs[2] on the same buffers: KS + MWU + Welch-t per position + Fisher window=5 (rank_hist_kernel)')
        del outs_all
        leg_done('all_tests')
    if headline_default and 'int16' in legs:
        # (b) the same rows as int16 milli-units, the format of real events (myRefBaseSignalAnnotation.py:1108): 844 B / position
        for b in blocks:
            b['q0'] = torch.empty(b['n'] * n0, dtype=torch.int16, device=dev); b['q1'] = torch.empty(b['n'] * n1, dtype=torch.int16, device=dev)
            det.synth_fill(b['q0'], SEED, b['lo_h'], b['n'], 0, n0, PLANT_PERIOD, PLANT_SHIFT)
            det.synth_fill(b['q1'], SEED, b['lo_h'], b['n'], 1, n1, PLANT_PERIOD, PLANT_SHIFT)
        det_q = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method=method, tests=tests, flags=det.flags)
        side['int16'] = side_leg(det_q, ('q0', 'q1'), blocks[0]['out'], False, method, d_gate[0],
                                 'the same generator as int16 milli-units (844 algorithmic bytes per position): ks_rank_kernel<16,16,i16>, packed v_pk_min/max_i16 sort')
        for b in blocks:
            del b['q0'], b['q1']
        leg_done('int16')
    if headline_default and 'rational_d' in legs and not rational_d:
        # (c) NMOD_FLAG_KS_RATIONAL_D: D as the correctly rounded rational (<= 2 ulp from ks_2samp's float form) — an opt-out no
        # reference-shaped entry point uses; rounds 1-3 quoted this rate as the headline
        det_r = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method=method, tests=tests, flags=L.FLAG_KS_RATIONAL_D | form_flags)
        side['rational_d'] = side_leg(det_r, ('sig0', 'sig1'), blocks[0]['out'], False, method, KS_D_RATIONAL_ABS,
                                      'flags = NMOD_FLAG_KS_RATIONAL_D (skips the float-form pass of D; gate 4.5e-16); the headline of BENCH_r01..r03')
        leg_done('rational_d')
    if simple and args.ties == 'few' and args.spread == 0 and args.dtype == 'f32' and args.config in ('ecoli', 'alltests') and 'real_ties' in legs:
        # (d) tie-heavy input (real NanoMod events are 3-decimal values): the buffers are refilled in place — last, nothing
        # after this leg sees the headline's rows
        for b in blocks:
            fill(b, 'real')
        side['real_ties'] = side_leg(det, ('sig0', 'sig1'), blocks[0]['out'], all_tests, method, d_gate[0],
                                     'the same generator on the 3-decimal grid (round(1000 x) / 1000 as float32): ties between and inside the groups as in real events')
        for b in blocks:
            fill(b, args.ties)
        leg_done('real_ties')
    if headline_default and ('real_spread' in legs or 'outliers' in legs):
        # (e) event-like rows: a signal level per position (+-3 units, both groups), reads spread sigma around it, 3-decimal grid —
        # most samples of a position tie with another one.  All three tests (what getKStest runs on every position) and KS + Stouffer
        # at sigma = 0.2 as float32 and as int16 milli-units (`real_spread_sweep`: sigma = 0.1 and 0.4 as well); `outliers`: the same
        # rows with 1 and 10 reads per 1 000 replaced by a uniform draw over +-5 units (mis-segmented events).  Each pass is checked
        # against the oracle and says which K1 form took its positions
        det_all = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method='fisher', tests=L.TEST_ALL, flags=form_flags)
        outs_all = det_all.alloc_outputs(blocks[0]['n'])
        det_q = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method=method, tests=tests, flags=det.flags)
        for b in blocks:
            b['q0'] = torch.empty(b['n'] * n0, dtype=torch.int16, device=dev); b['q1'] = torch.empty(b['n'] * n1, dtype=torch.int16, device=dev)
        rs = {'note': 'nmod_synth_fill_events: level(position) in +-3 units shared by both groups, reads spread sigma around it, values on the '
                      '3-decimal grid of stored events (myRefBaseSignalAnnotation.py:1108); the same 4.6 M x 200 v 200 positions'}
        ol = {'note': 'the event-like rows at sigma = 0.2 with N reads per 1 000 replaced by a uniform draw over +-5 units (the clip range of the raw '
                      'normalisation, myRefBaseSignalAnnotation.py:251-259): what a counting form does with samples outside its window'}
        plan = []
        if 'real_spread' in legs:
            plan += [(sg, 0) for sg in ((200, 100, 400) if 'real_spread_sweep' in legs else (200,))]
        if 'outliers' in legs:
            plan += [(200, 1), (200, 10)]
        for sg, outl in plan:
            for b in blocks:
                fill(b, 'few', sg, outliers=outl); fill(b, 'few', sg, ('q0', 'q1'), outliers=outl)
            tag = 'sigma_0.%d' % (sg // 100) if not outl else '%d_permille' % outl
            dst = ol if outl else rs
            what = 'sigma = %.1f' % (sg / 1000) + (', %d per mille outliers' % outl if outl else '')
            dst['all_tests_f32_' + tag] = side_leg(det_all, ('sig0', 'sig1'), outs_all, True, 'fisher', 0.0, 'all three tests + Fisher, float32 rows, ' + what)
            dst['all_tests_i16_' + tag] = side_leg(det_all, ('q0', 'q1'), outs_all, True, 'fisher', 0.0, 'all three tests + Fisher, int16 milli-unit rows, ' + what)
            if sg == 200 and not outl:
                dst['ks_f32_' + tag] = side_leg(det, ('sig0', 'sig1'), blocks[0]['out'], False, method, d_gate[0], 'KS + Stouffer, float32 rows, sigma = 0.2')
                dst['ks_i16_' + tag] = side_leg(det_q, ('q0', 'q1'), blocks[0]['out'], False, method, d_gate[0], 'KS + Stouffer, int16 milli-unit rows, sigma = 0.2')
        if 'real_spread' in legs:
            side['real_spread'] = rs
        if 'outliers' in legs:
            side['outliers'] = ol
        del outs_all
        for b in blocks:
            del b['q0'], b['q1']
            fill(b, args.ties)
        leg_done('real_spread+outliers')
    if len(side):
        step(False); state.wait(); torch.cuda.synchronize()     # the headline's outputs are back for what follows

    # ---- the other BASELINE configurations at their per-GPU size, a few timed steps each (configs[3]: one GPU's share of chr20;
    # configs[4]: the ragged stress, KS + Stouffer and all three tests): every BASELINE config gets a number in the default run
    ragged_cache = {}

    def preset_rows(name, P):
        """sizes / offsets of a preset's rows (host + device), made once per run: 10 M ragged sizes are a second of numpy each"""
        if (name, P) not in ragged_cache:
            hoff, doff = [None, None], [None, None]
            for g in (0, 1):
                sz = ragged_sizes(SEED, 0, P, g)
                hoff[g] = np.zeros(P + 1, np.int64)
                np.cumsum(sz, out=hoff[g][1:])
                doff[g] = torch.from_numpy(hoff[g]).to(dev)
            ragged_cache[(name, P)] = (hoff, doff)
        return ragged_cache[(name, P)]

    def preset_leg(name, leg_all, steps, spread=0, i16=False, positions=None, outliers=0):
        pz = PRESETS[name]
        csr_ = pz['layout'] == 'csr'
        P, m0, m1 = positions or pz['positions'], pz['n0'], pz['n1']
        leg_method = 'fisher' if leg_all else 'stouffer'
        tdt = torch.int16 if i16 else torch.float32
        shift_m = int(round(PLANT_SHIFT * 1000))
        det_x = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method=leg_method, tests=L.TEST_ALL if leg_all else L.TEST_KS, flags=form_flags)
        rid_x = torch.zeros(P, dtype=torch.int32, device=dev)
        sig, hoff, doff = [None, None], [None, None], [None, None]
        if csr_:
            hoff, doff = preset_rows(name, P)
        for g in (0, 1):
            if csr_:
                sig[g] = torch.empty(int(hoff[g][-1]), dtype=tdt, device=dev)
                if spread:
                    det_x.synth_fill_events(sig[g], SEED, 0, P, g, n_per_pos=0, off=doff[g], plant_period=PLANT_PERIOD, plant_shift_milli=shift_m, spread_milli=spread,
                                            outlier_permille=outliers)
                else:
                    det_x.synth_fill_csr(sig[g], SEED, 0, doff[g], g, PLANT_PERIOD, PLANT_SHIFT)
            else:
                sig[g] = torch.empty(P * (m0, m1)[g], dtype=tdt, device=dev)
                if spread:
                    det_x.synth_fill_events(sig[g], SEED, 0, P, g, n_per_pos=(m0, m1)[g], plant_period=PLANT_PERIOD, plant_shift_milli=shift_m, spread_milli=spread,
                                            outlier_permille=outliers)
                else:
                    det_x.synth_fill(sig[g], SEED, 0, P, g, (m0, m1)[g], PLANT_PERIOD, PLANT_SHIFT)
        outs = det_x.alloc_outputs(P)

        def run_once():
            if csr_:
                det_x.run(sig[0], sig[1], rid_x, off0=doff[0], off1=doff[1], max_n0=RAGGED_CLIP0[1], max_n1=RAGGED_CLIP1[1], out=outs)
            else:
                det_x.run(sig[0], sig[1], rid_x, stride0=m0, stride1=m1, npos=P, out=outs)
        run_once(); torch.cuda.synchronize()
        share = form_share(det_x.dispatch_stats())
        vn = 5_000 if csr_ else 20_000
        if csr_:
            o0, o1 = hoff[0][:vn + 1], hoff[1][:vn + 1]
        else:
            o0 = np.arange(0, (vn + 1) * m0, m0, dtype=np.int64); o1 = np.arange(0, (vn + 1) * m1, m1, dtype=np.int64)
        rows = (sig[0][:int(o0[-1])].cpu().numpy(), o0, sig[1][:int(o1[-1])].cpu().numpy(), o1)
        v = verify_against_oracle(rows, vn, outs, leg_all, leg_method, 0.0)
        run_once()
        tm = nm.EventTimer(256)
        det_x.timer = tm
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_once()
        barrier()
        el = time.perf_counter() - t0
        det_x.timer = None
        k1, kn = tm.read(L.KERNEL_RANK_STATS); k2, _ = tm.read(L.KERNEL_FINALIZE); k3, _ = tm.read(L.KERNEL_COMBINE)
        mean0_, mean1_ = sig[0].numel() / P, sig[1].numel() / P
        algo_leg = algorithmic_bytes(mean0_, mean1_, 2 if i16 else 4, 4 if leg_all else 2, csr_)
        gbs = algo_leg * P / ((k1 + k2 + k3) / steps * 1e-3) / 1e9
        if not v['ok']:
            verify['ok'] = False
            print('bench.py: verification of the %s leg FAILED: %r' % (name, v), file=sys.stderr)
        return {'value': P * steps / el, 'unit': 'positions/s', 'steps': steps, 'ms_per_step': el / steps * 1e3, 'positions': P,
                'workload': '%s, %s, %s%s%s' % (pz['name'], 'KS + MWU + Welch-t + Fisher' if leg_all else 'KS + weighted Stouffer', 'int16 milli-units' if i16 else 'float32',
                                                (', event-like rows (sigma = %.1f)' % (spread / 1000)) if spread else '',
                                                (', %d per mille outliers' % outliers) if outliers else ''),
                'mean_reads': [mean0_, mean1_], 'k1_ms_per_step': k1 / steps, 'algorithmic_bytes_per_position': algo_leg,
                'achieved_GBps': gbs, 'roofline_frac': gbs / HBM_PEAK_GBS, 'form_share': share, 'verify': v}

    # ---- the host-resident entry on the same rows (NMOD_MEM_HOST): PCIe-bound, its own roofline
    host_path = None
    if world == 1 and not csr and not args.no_host_path and chunks == 1 and not args.force_collective and ('host_path' in legs or not headline_default):
        host_path = host_path_leg(nm, torch, local_rank, blocks, n0, n1, NB, WDIF, method, tests,
                                  want_i16=(args.dtype == 'f32'), ref_out=blocks[0]['out'], flags=det.flags)
        bad = [k for k, v in host_path.items() if isinstance(v, dict) and (v.get('equals_device_resident_pass') is False or v.get('equals_int16_pass') is False)]
        if bad:
            verify['ok'] = False
            print('bench.py: the host-resident entry differs from the device-resident pass: %r' % bad, file=sys.stderr)
        if rank == 0:
            emit_line(json.dumps({'side_leg': 'host_path', 'record': host_path}))
        leg_done('host_path')

    # ---- the function-level drop-in on the reference's dict shape (host glue + PCIe + kernels + table), a tenth of E. coli
    drop_in = None
    if headline_default and not args.no_host_path and not args.positions and ('drop_in' in legs or 'drop_in_200' in legs):
        # (`drop_in`: 20 v 20 reads, per-position arrays and lists of numpy.float64; `drop_in_200`: only the leg below)
        drop_in = drop_in_leg(nm, local_rank) if 'drop_in' in legs else {}
        # ... and at the north-star coverage: 460 000 positions x 200 v 200 reads (1.5 GB of float64 rows as per-position arrays),
        # with the stages of mtest2 timed one by one
        drop_in['at_200v200'] = drop_in_leg(nm, local_rank, 460_000, 200, shapes=('arrays',), split=True, event_like=True)
        flat = [v for v in drop_in.values() if isinstance(v, dict) and 'verify_ok' in v] + \
               [v for v in drop_in['at_200v200'].values() if isinstance(v, dict) and 'verify_ok' in v]
        if not all(v['verify_ok'] for v in flat):
            verify['ok'] = False
            print('bench.py: the drop-in mtest2 leg differs from the oracle: %r' % drop_in, file=sys.stderr)
        if rank == 0:
            emit_line(json.dumps({'side_leg': 'drop_in_mtest2', 'record': drop_in}))
        leg_done('drop_in')

    # (after the host-resident leg: freeing the presets' 79 GB of device memory slows the PCIe copies that follow for a while —
    # 0.77 instead of 0.95 of the pinned rate when the order is reversed)
    if headline_default and not args.positions and 'presets' in legs:
        side['chr20_share'] = preset_leg('chr20', False, 5)
        torch.cuda.empty_cache()
        side['ragged'] = preset_leg('ragged', False, 3)
        torch.cuda.empty_cache()
        leg_done('presets')
        # configs[4] on event-like int16 rows (sigma = 0.2), all three tests — what getKStest computes on stored events at real, ragged
        # coverage: the counting form for any coverage where the device-side probe accepts a class — clean and with 1 / 10 per mille outliers
        if 'outliers' in legs or 'presets_event' in legs:
            ev = {'note': 'nmod_synth_fill_events, sigma = 0.2, on the ragged preset (~1 131 v ~57), all three tests + Fisher; N per mille of the reads replaced by outliers over +-5 units'}
            for outl in (0, 1, 10):
                ev['ragged_i16_%d_permille' % outl] = preset_leg('ragged', True, 3, spread=200, i16=True, outliers=outl)
                torch.cuda.empty_cache()
            side['ragged_event_outliers'] = ev
            leg_done('ragged_event_outliers')
        if 'presets_event' in legs:
            side['ragged_all_tests'] = preset_leg('ragged', True, 3)
            torch.cuda.empty_cache()
            ev = {'note': 'nmod_synth_fill_events, sigma = 0.2; the ragged preset (~1 131 v ~57) and the chr20 shape (500 v 500); all three tests + Fisher, and (_ks_) KS + Stouffer; '
                          '_1pm / _10pm: with 1 / 10 per mille outliers'}
            for nm_, kw in (('ragged_f32', dict(name='ragged', i16=False)), ('ragged_f32_1pm', dict(name='ragged', i16=False, outliers=1)),
                            ('ragged_f32_10pm', dict(name='ragged', i16=False, outliers=10)),
                            ('chr20_i16', dict(name='chr20', i16=True)), ('chr20_f32', dict(name='chr20', i16=False)),
                            # ... and the presets' own mask, KS + Stouffer (the form without the tie term and the moments)
                            ('ragged_ks_i16', dict(name='ragged', i16=True, ks=True)), ('chr20_ks_i16', dict(name='chr20', i16=True, ks=True)),
                            ('chr20_ks_f32', dict(name='chr20', i16=False, ks=True))):
                ev[nm_] = preset_leg(kw['name'], not kw.get('ks', False), 3, spread=200, i16=kw['i16'], outliers=kw.get('outliers', 0))
                torch.cuda.empty_cache()
            side['real_spread_presets'] = ev
            leg_done('presets_event')
    ragged_cache.clear()

    line = None
    if rank == 0:
        value = total * args.steps / elapsed
        launches = max(k1_n, 1)
        steps_launches = args.steps * chunks                                  # blocks timed (one event pair per block and kernel)
        k1_per_block_s = k1_ms / steps_launches * 1e-3
        path_per_block_s = (k1_ms + k2_ms + k3_ms) / steps_launches * 1e-3
        pos_per_launch = n_local / chunks
        k_out = 4 if all_tests else 2
        mean0 = (sum(b['sig0'].numel() for b in blocks) / n_local)
        mean1 = (sum(b['sig1'].numel() for b in blocks) / n_local)
        algo = algorithmic_bytes(mean0, mean1, sample_bytes, k_out, csr)
        k1_bytes = sample_bytes * (mean0 + mean1) + (16 if csr else 0) + (4 if not all_tests else 8 + 8 + 32 + 8)   # K1's own reads + writes
        achieved = algo * pos_per_launch / path_per_block_s / 1e9 if path_per_block_s > 0 else 0.0
        k1_achieved = k1_bytes * pos_per_launch / k1_per_block_s / 1e9 if k1_per_block_s > 0 else 0.0
        prm = L.make_params(dtype=L.DTYPE_F32 if args.dtype == 'f32' else L.DTYPE_I16_MILLI, tests=tests,
                            method=L.METHOD_BY_NAME[method])
        import ctypes
        kbuf = ctypes.create_string_buffer(128)
        L.check(L.load().nmod_describe_dispatch(ctypes.byref(prm), n0, n1, kbuf, 128), 'nmod_describe_dispatch')
        shape = 'ragged' if csr else '%dv%d' % (n0, n1)
        key = '%s_%s_%s_%d%s' % ('all' if all_tests else 'ks', args.dtype, shape, int(pos_per_launch), ('_realties' if args.ties == 'real' else '') + ('_spread%d' % args.spread if args.spread else '') + ('_outl%d' % args.outliers if args.outliers else '') + ('_rationald' if rational_d else ''))
        rec = profile_record(L.LIB_PATH, key) or {}
        tests_txt = 'KS + MWU + Welch-t + Fisher window=%d' % (2 * NB + 1) if all_tests else 'KS + weighted Stouffer window=%d' % (2 * NB + 1)
        reads_txt = ('n0 ~ LogNormal(ln 1000, 0.5) in [5, 4000], n1 ~ LogNormal(ln 50, 0.5) in [5, 400] (means %.0f v %.0f), CSR'
                     % (mean0, mean1)) if csr else '%d v %d reads/position' % (n0, n1)
        line = {
            'metric': 'genomic positions/sec (%s)' % ('KS + MWU + Welch-t + Fisher' if all_tests else 'KS + Stouffer'),
            'value': value, 'unit': 'positions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
            'vs_baseline': None, 'dtype': '%s keys / f64 p-values' % args.dtype,
            'data': ('synthetic event-like rows (nmod_synth_fill_events: a level per position in +-3 units shared by both groups, reads spread '
                     '%.1f units around it — counter-based Irwin-Hall(4), support +-3.46 sigma — on the 3-decimal grid of stored events: most '
                     'samples of a position tie; +0.8 shift planted in group 2 every 10 000 positions)' % (args.spread / 1000)) if args.spread else
                    'synthetic (counter-based Irwin-Hall(4) on a 262 141-value grid, support +-3.46 sigma, unit variance — '
                    'a stand-in for N(0,1); %s; +0.8 shift planted in group 2 every 10 000 positions)'
                    % ('~1 tie per position' if args.ties == 'few' and args.dtype == 'f32' else 'on the 3-decimal grid of real events: ~11 ties between the groups per 200 v 200 position'),
            'config': {'workload': '%s%s: %d positions in total (%d per GPU), %s, %s'
                                   % (preset['name'], ' x %d' % world if (world > 1 and not args.strong) else '', total, total // world, reads_txt, tests_txt),
                       'preset': args.config, 'positions_total': total, 'positions_per_gpu': total // world, 'n0': n0, 'n1': n1,
                       'layout': 'csr' if csr else 'fixed stride', 'neighborPvalues': NB, 'WeightsDif': WDIF, 'ties': args.ties, 'spread_milli': args.spread,
                       'outlier_permille': args.outliers, 'flags': det.flags,
                       'parallelism_short': ('block-cyclic position shards x%d, %d rounds, +-%d halo, async RCCL all-gather per round' % (world, chunks, NB)) if gather else 'one GPU, no collective',
                       'ks_d': ('exact rational max|c0 n1 - c1 n0| / (n0 n1), correctly rounded (NMOD_FLAG_KS_RATIONAL_D; <= 2 ulp from '
                                'ks_2samp\'s float form, gate 4.5e-16)') if rational_d else
                               'ks_2samp\'s float form bit for bit (library default, flags = 0)',
                       'rccl_ranks': dist.get_world_size() if dist is not None else 0,
                       'backend': dist.get_backend() if dist is not None else None,
                       'parallelism': ('block-cyclic position sharding x%d, %d rounds of %d-position blocks, +-%d halo recomputed; '
                                       'per round one RCCL all_gather_into_tensor per track (ks_p, comb_p), issued async behind '
                                       'the round\'s kernels' % (world, chunks, B, NB)) if gather else
                                      'one GPU, one block, no collective'},
            'allgather_exposed_ms': None if exposed['per_rank_ms'] is None else
                                    {'per_rank': exposed['per_rank_ms'], 'max': max(exposed['per_rank_ms']), 'steps': args.steps,
                                     'note': 'HIP-event time between the last round\'s kernels and the end of the last outstanding all-gather on each '
                                             'rank\'s compute stream, after the K timed steps: the part of the gather that no kernel hides'},
            'compute_only': {'value': total * args.steps / compute_elapsed, 'unit': 'positions/s',
                             'ms_per_step': compute_elapsed / args.steps * 1e3,
                             'note': 'the same K steps without the all-gather (tracks stay sharded)' if gather else
                                     'identical to value: one rank has nothing to gather'},
            'roofline': {'bound': 'valu-issue', 'yardstick': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': rec.get('hbm_bytes_per_launch'), 'traffic_source': rec.get('source'),
                         'note': 'achieved / peak / frac are against HBM, the mandated yard-stick (sort / search / scan, no MFMA); what '
                                 'bounds the kernel is VALU instruction issue (see `valu`): traffic ~ 1.0 x algorithmic, ~0.9 of the issue '
                                 'slots taken, clock held near 2.2 GHz by the power limit.  `floor_instr_per_position` is the budget of this '
                                 'design (profiles/HISTORY.md B.3: sort of the smaller group + one binary search per sample + prefix-sum evaluation)',
                         'floor_instr_per_position': FLOOR_INSTR.get(('all' if all_tests else 'ks', n0, n1)),
                         'definition': 'SURVEY.md 8(d) bytes/position x positions per launch / HIP-event time of K1 + K2 + K3 of that launch',
                         'algorithmic_bytes_per_position': algo, 'positions_per_launch': pos_per_launch, 'profile_key': key,
                         'path_avg_ms': path_per_block_s * 1e3,
                         'kernel': kbuf.value.decode() if not csr else 'size-class launches (ragged); at the median sizes: ' + kbuf.value.decode(),
                         'kernel_avg_ms': k1_per_block_s * 1e3, 'launches_timed': k1_n,
                         'dominant_kernel_only': {'bytes_per_position': k1_bytes, 'achieved': k1_achieved,
                                                  'frac': k1_achieved / HBM_PEAK_GBS,
                                                  'note': 'bytes K1 itself reads and writes / K1 time (all size-class launches of a block)'},
                         'frac_of_measured_copy': achieved / copy_gbs if copy_gbs else None,
                         'measured_copy_GBps': copy_gbs, 'guide_copy_GBps': HBM_COPY_GUIDE_GBS,
                         'other_kernels_avg_ms': {'finalize': k2_ms / steps_launches, 'combine': k3_ms / steps_launches}},
            'valu': {'instr_per_position': rec.get('valu_instr_per_position'), 'issue_util': rec.get('valu_issue_util'),
                     'source': rec.get('source') if rec.get('valu_instr_per_position') is not None else None,
                     'note': 'rocprofv3 SQ_INSTS_VALU / positions and SQ_ACTIVE_INST_VALU x 4 / SIMD cycles of the K1 kernel, taken '
                             'with this library binary (null: this binary has not been profiled)'},
            'form_share': headline_share,
            'verify': verify,
            'build_info': L.load().nmod_build_info().decode(),
        }
        line['data_short'] = ('synthetic event-like rows, sigma %.1f, %d per mille outliers, 3-decimal grid' % (args.spread / 1000, args.outliers)) if args.spread else \
                             ('synthetic Irwin-Hall(4) stand-in for N(0,1)%s, +0.8 shift planted every 10 000 positions' % (' on the 3-decimal grid' if args.ties == 'real' else ''))
        line['build_info_short'] = short_build_info(line['build_info'])
        try:
            line['lib_sha16'] = hashlib.sha256(open(L.LIB_PATH, 'rb').read()).hexdigest()[:16]
        except OSError:
            line['lib_sha16'] = None
        if not args.no_cpu and world == 1:           # the CPU baseline is an N=1 figure
            line['cpu_baseline'] = cpu_baseline(cpu_rows, '%s, %s' % (reads_txt, tests_txt), method, 7 if all_tests else 1, usable_cpus(), refpy=refpy)
            leg_done('cpu_baseline')
        line['leg_seconds'] = leg_seconds
    ok = torch.tensor([1 if (rank != 0 or verify['ok']) else 0], device=dev)
    if dist is not None:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so that the JSON line is the last line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        # the verbose record and every side leg go to the side file (each leg was a stdout line of its own when it finished); the LAST
        # stdout line is the compact record the driver parses
        full = dict(line)
        full['side_legs'] = dict(side)
        if host_path is not None:
            full['host_path'] = host_path
        if drop_in is not None:
            full['drop_in_mtest2'] = drop_in
        side_file = args.side_file
        try:
            with open(side_file, 'w') as f:
                json.dump(full, f, indent=1)
        except OSError as e:
            print('bench.py: could not write %s: %s' % (side_file, e), file=sys.stderr)
            side_file = None
        rel = os.path.relpath(side_file, ROOT) if side_file else None
        rec_, text = compact_record(line, side, host_path, drop_in, rel)
        emit_line(text)
    if int(ok.item()) == 0:
        sys.exit(3)                                  # a numerically wrong build must not look like a benchmark record


if __name__ == '__main__':
    main()
