#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over this job's synthetic genome: K1 rank statistics + K2
p-values + K3 window combine on every rank's positions and — for N > 1 — the RCCL all-gather that
reassembles the per-base KS-p and combined-p tracks on every rank (BASELINE.json north_star).
Inputs are resident in HBM when the timed region starts.

Workload at N=1: BASELINE.json configs[1] — E. coli 4.6 Mb, 200 v 200 reads/position, KS + weighted
Stouffer (window 5), float32 signals.  N > 1, weak scaling (default): an N x 4.6 M position genome,
4.6 M positions per rank; --strong: 4.6 M positions in total.  Positions are dealt block-cyclic
(nanomod_amd/sharding.py: `chunks` rounds of N equal blocks, each block with a +-nb halo of recomputed
neighbours), so the all-gather of round c lands as one contiguous piece of the full track and runs on
RCCL's stream beside the kernels of round c+1.  Two figures are timed: `value` = kernels + all-gather
(for N=1 there is nothing to gather), and `compute_only` = the kernels alone.

Before the warm-up one untimed pass is checked against the CPU oracle (the C restatement) on a bounded
sample of the same input; the result goes into `verify`.  The oracle is also the `cpu_baseline`.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P_ECOLI = 4_600_000
NB = 2
WDIF = 2.0
SEED = 20240601
PLANT_PERIOD = 10000
PLANT_SHIFT = 0.8
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GUIDE_GBS = 6290.0    # same guide: measured float4 copy


def algorithmic_bytes(n0, n1, sample_bytes, k_out):
    """s*(n0+n1) [samples] + 16 [CSR offsets; 0 for the fixed-stride layout used here] + 4 [run id]
    + 16*k_out [(stat, p) fp64 pairs written] + 8 [own-p re-read by the combine]."""
    return sample_bytes * (n0 + n1) + 0 + 4 + 16 * k_out + 8


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


def oracle_run(a, b, npos, n0, n1, method, tests, threads):
    """The checker: oracle/nanomod_oracle.c on the first `npos` rows of a, b.  Returns (outputs, seconds)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import oracle_c
    off0 = np.arange(0, (npos + 1) * n0, n0, dtype=np.int64)
    off1 = np.arange(0, (npos + 1) * n1, n1, dtype=np.int64)
    t0 = time.perf_counter()
    out = oracle_c.detect_batch(a[:npos].reshape(-1), off0, b[:npos].reshape(-1), off1, np.zeros(npos, np.int32),
                                NB, WDIF, method, tests=tests, threads=threads)
    return out, time.perf_counter() - t0


def cpu_baseline(a, b, n0, n1, method, tests, threads, target_seconds=12.0):
    """The oracle timed on this box's host cores on a bounded sample of the same workload (the first rows of rank 0's
    device-resident input, copied back): all usable cores, and one core."""
    import numpy as np
    cap = a.shape[0]
    oracle_run(a, b, min(2000, cap), n0, n1, method, tests, threads)                 # thread-pool warm-up
    probe = min(20000, cap)
    rate = probe / oracle_run(a, b, probe, n0, n1, method, tests, threads)[1]
    sample = int(min(cap, max(probe, rate * target_seconds)))
    reps = max(1, int(round(rate * target_seconds / sample)))
    dt = sum(oracle_run(a, b, sample, n0, n1, method, tests, threads)[1] for _ in range(reps))
    one = int(min(cap, max(2000, rate / max(threads, 1) * 4.0)))                      # ~4 s on one core
    dt1 = oracle_run(a, b, one, n0, n1, method, tests, 1)[1]
    # the reference's own shape of the computation — one scipy-style call sequence per position in Python, one core,
    # all three tests as getKStest always computes them — on a small sample, for scale
    import nanomod_oracle as orc
    npy = min(300, cap)
    scale = 1e-3 if a.dtype == np.int16 else 1.0
    t0 = time.perf_counter()
    ksp = [orc.getKStest(a[i].astype(np.float64) * scale, b[i].astype(np.float64) * scale)[2][1] for i in range(npy)]
    orc.combine_track(np.zeros(npy), np.array(ksp), np.zeros(npy, np.int32), NB, WDIF,
                      orc.METHOD_STOUFFER if method == 'stouffer' else orc.METHOD_FISHER)
    dpy = time.perf_counter() - t0
    what = 'all three tests + Fisher' if tests == 7 else 'KS + Stouffer window 5'
    return {'value': sample * reps / dt, 'unit': 'positions/s', 'cores': threads, 'kind': 'port', 'cpu_model': cpu_model(),
            'sample': 'first %d positions of the same workload (%d v %d, %s) x %d passes, oracle/nanomod_oracle.c '
                      'with OpenMP on %d threads (cgroup CPU quota), %.1f s' % (sample, n0, n1, what, reps, threads, dt),
            'one_core': {'value': one / dt1, 'unit': 'positions/s', 'cores': 1,
                         'sample': 'first %d positions, same library, 1 thread, %.1f s' % (one, dt1)},
            'reference_shaped_python': {'value': npy / dpy, 'unit': 'positions/s', 'cores': 1,
                                        'sample': 'oracle/nanomod_oracle.py getKStest + combine per position on the first %d '
                                                  'positions (the reference computes MWU, Welch and KS for every position)' % npy}}


def measured_traffic(lib_path, key):
    """HBM bytes per K1 launch from a rocprofv3 PMC run (tools/pmc_traffic.py writes profiles/traffic.json), used only
    if it was taken with the library binary that is running now."""
    try:
        sha = hashlib.sha256(open(lib_path, 'rb').read()).hexdigest()[:16]
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        ent = rec.get(key)
        if ent and ent.get('lib_sha16') == sha:
            return ent['hbm_bytes_per_launch'], ent.get('source')
    except Exception:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--positions', type=int, default=P_ECOLI, help='positions per GPU (weak) or in total (--strong)')
    ap.add_argument('--n0', type=int, default=200)
    ap.add_argument('--n1', type=int, default=200)
    ap.add_argument('--dtype', choices=('f32', 'i16'), default='f32', help='sample dtype in HBM (i16 = milli-units)')
    ap.add_argument('--all-tests', action='store_true', help='BASELINE.json configs[2]: KS + MWU + Welch-t + Fisher (not the headline metric)')
    ap.add_argument('--strong', action='store_true', help='fixed total size: --positions in total, split over the ranks')
    ap.add_argument('--chunks', type=int, default=0, help='rounds of the block-cyclic pipeline (default 4 for N>1, 1 for N=1)')
    ap.add_argument('--force-collective', action='store_true', help='N=1: initialise RCCL with one rank and issue the all-gather anyway')
    ap.add_argument('--cpu-sample', type=int, default=0, help='cap on positions for the CPU baseline / verification (0 = 1 M)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline (the verification still runs)')
    args = ap.parse_args()

    import numpy as np
    import torch
    import nanomod_amd as nm
    from nanomod_amd import sharding
    L = nm._lib

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run' % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))   # "nccl" = RCCL on ROCm
    gather = dist is not None

    n0, n1 = args.n0, args.n1
    tdtype = torch.float32 if args.dtype == 'f32' else torch.int16
    sample_bytes = 4 if args.dtype == 'f32' else 2
    chunks = args.chunks or (4 if world > 1 else 1)
    total = args.positions if args.strong else args.positions * world
    B = sharding.cyclic_block_len(total, world, chunks)
    total = B * world * chunks                     # the synthetic genome is padded to whole blocks
    method = 'fisher' if args.all_tests else 'stouffer'
    tests = L.TEST_ALL if args.all_tests else L.TEST_KS
    det = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method=method, tests=tests)

    # this rank's blocks (+ halo), generated on the device from global position counters
    blocks = []
    for c in range(chunks):
        lo, hi = sharding.cyclic_block(total, world, rank, chunks, c)
        lo_h, hi_h = sharding.halo_bounds(lo, hi, NB, total)
        n = hi_h - lo_h
        s0 = torch.empty(n * n0, dtype=tdtype, device=dev)
        s1 = torch.empty(n * n1, dtype=tdtype, device=dev)
        det.synth_fill(s0, SEED, lo_h, n, 0, n0, PLANT_PERIOD, PLANT_SHIFT)
        det.synth_fill(s1, SEED, lo_h, n, 1, n1, PLANT_PERIOD, PLANT_SHIFT)
        blocks.append({'lo_h': lo_h, 'hi_h': hi_h, 'n': n, 'sig0': s0, 'sig1': s1,
                       'rid': torch.zeros(n, dtype=torch.int32, device=dev),       # one contiguous run
                       'out': det.alloc_outputs(n)})
    n_local = sum(b['hi_h'] - b['lo_h'] for b in blocks)
    state = sharding.PipelinedGather(total, world, chunks, ('ks_p', 'comb_p'), dev)

    def compute(c, lo_h, hi_h):
        b = blocks[c]
        assert (lo_h, hi_h) == (b['lo_h'], b['hi_h'])
        return det.run(b['sig0'], b['sig1'], b['rid'], stride0=n0, stride1=n1, npos=b['n'], out=b['out'])

    def step(with_gather):
        sharding.pipelined_detect(compute, state, NB, gather=with_gather, force_collective=args.force_collective)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(with_gather, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(with_gather)
        state.wait()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- one untimed pass, checked on rank 0 against the CPU oracle on a bounded sample of the same input
    step(gather)
    state.wait()
    torch.cuda.synchronize()
    verify = None
    cpu_rows = None
    if rank == 0:
        cap = min(args.cpu_sample or 1_000_000, blocks[0]['n'])
        a = blocks[0]['sig0'][:cap * n0].cpu().numpy().reshape(cap, n0)
        b = blocks[0]['sig1'][:cap * n1].cpu().numpy().reshape(cap, n1)
        cpu_rows = (a, b)
        vn = min(cap, 200_000)
        exp, _ = oracle_run(a, b, vn, n0, n1, method, 7 if args.all_tests else 1, usable_cpus())
        verify = {'positions': vn, 'against': 'oracle/nanomod_oracle.c on the first positions of rank 0, same input'}
        names = ['ks_d', 'ks_p', 'comb_st', 'comb_p'] + (['mwu_u', 'mwu_p', 't_t', 't_p'] if args.all_tests else [])
        inner = slice(0, vn - NB)                   # the sample's last nb positions see neighbours the oracle run did not
        ok = True
        for k in names:
            g = blocks[0]['out'][k][:vn].cpu().numpy()[inner]
            e = exp[k][inner]
            fin = np.isfinite(e)
            same_special = bool(np.array_equal(g[~fin], e[~fin], equal_nan=True))
            rel = float(np.max(np.abs(g[fin] - e[fin]) / np.maximum(np.abs(e[fin]), 1e-300))) if fin.any() else 0.0
            ab = float(np.max(np.abs(g[fin] - e[fin]))) if fin.any() else 0.0
            verify['max_rel_err_' + k] = rel
            verify['max_abs_err_' + k] = ab
            # the gates of tests/helpers.py: p-values 1e-9 relative and 1e-6 absolute (north_star), statistics 1e-9
            # relative + 1e-12 absolute, D within 4.5e-16 (KS-only mode: the exact rational; bit-exact with all tests)
            err = np.abs(g[fin] - e[fin])
            if k.endswith('_p'):
                good = bool(np.all(err <= 1e-9 * np.abs(e[fin]) + 1e-300)) and ab <= 1e-6
            elif k == 'ks_d':
                good = ab <= 4.5e-16
            else:
                good = bool(np.all(err <= 1e-9 * np.abs(e[fin]) + 1e-12))
            ok = ok and same_special and good
        if gather:                                   # the gathered track holds rank 0's first block at its natural place
            full = state.result()
            ok = ok and bool(torch.equal(full['ks_p'][blocks[0]['lo_h']:blocks[0]['lo_h'] + vn], blocks[0]['out']['ks_p'][:vn]))
        verify['ok'] = bool(ok)
        if not ok:
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

    if rank == 0:
        value = total * args.steps / elapsed
        launches = max(k1_n, 1)
        k1_avg_s = k1_ms / launches * 1e-3
        path_avg_s = (k1_ms + k2_ms + k3_ms) / launches * 1e-3
        pos_per_launch = n_local / chunks
        k_out = 4 if args.all_tests else 2
        algo = algorithmic_bytes(n0, n1, sample_bytes, k_out)
        k1_bytes = sample_bytes * (n0 + n1) + (4 if not args.all_tests else 8 + 8 + 32 + 8)   # K1's own reads + writes
        achieved = algo * pos_per_launch / path_avg_s / 1e9 if path_avg_s > 0 else 0.0
        k1_achieved = k1_bytes * pos_per_launch / k1_avg_s / 1e9 if k1_avg_s > 0 else 0.0
        prm = L.make_params(dtype=L.DTYPE_F32 if args.dtype == 'f32' else L.DTYPE_I16_MILLI, tests=tests,
                            method=L.METHOD_BY_NAME[method])
        import ctypes
        kbuf = ctypes.create_string_buffer(96)
        L.check(L.load().nmod_describe_dispatch(ctypes.byref(prm), n0, n1, kbuf, 96), 'nmod_describe_dispatch')
        traffic, traffic_src = measured_traffic(L.LIB_PATH, '%s_%s_%dv%d_%d' % ('all' if args.all_tests else 'ks', args.dtype, n0, n1, int(pos_per_launch)))
        what = 'KS + MWU + Welch-t + Fisher window=%d (BASELINE.json configs[2])' % (2 * NB + 1) if args.all_tests \
            else 'KS + weighted Stouffer window=%d (BASELINE.json configs[1])' % (2 * NB + 1)
        line = {
            'metric': 'genomic positions/sec (%s)' % ('KS + MWU + Welch-t + Fisher' if args.all_tests else 'KS + Stouffer'),
            'value': value, 'unit': 'positions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
            'vs_baseline': None, 'dtype': '%s keys / f64 p-values' % args.dtype,
            'data': 'synthetic (counter-based Irwin-Hall(4) on a 262 141-value grid, support +-3.46 sigma, unit variance — '
                    'a stand-in for N(0,1) that keeps ~1 tie per position; +0.8 shift planted in group 2 every 10 000 positions)',
            'config': {'workload': 'E. coli 4.6 Mb x %d: %d positions in total, %d v %d reads/position, %s'
                                   % (world if not args.strong else 1, total, n0, n1, what),
                       'positions_total': total, 'positions_per_gpu': total // world, 'n0': n0, 'n1': n1,
                       'neighborPvalues': NB, 'WeightsDif': WDIF,
                       'parallelism': ('block-cyclic position sharding x%d, %d rounds of %d-position blocks, +-%d halo recomputed; '
                                       'per round one RCCL all_gather_into_tensor per track (ks_p, comb_p), issued async behind '
                                       'the round\'s kernels' % (world, chunks, B, NB)) if gather else
                                      'one GPU, one block, no collective'},
            'compute_only': {'value': total * args.steps / compute_elapsed, 'unit': 'positions/s',
                             'ms_per_step': compute_elapsed / args.steps * 1e3,
                             'note': 'the same K steps without the all-gather (tracks stay sharded)' if gather else
                                     'identical to value: one rank has nothing to gather'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'definition': 'SURVEY.md 8(d) bytes/position (fixed stride: no CSR offsets) x positions per launch / '
                                       'HIP-event time of K1 + K2 + K3 of that launch',
                         'algorithmic_bytes_per_position': algo, 'positions_per_launch': pos_per_launch,
                         'path_avg_ms': path_avg_s * 1e3,
                         'kernel': kbuf.value.decode(), 'kernel_avg_ms': k1_avg_s * 1e3, 'launches_timed': k1_n,
                         'dominant_kernel_only': {'bytes_per_position': k1_bytes, 'achieved': k1_achieved,
                                                  'frac': k1_achieved / HBM_PEAK_GBS,
                                                  'note': 'bytes K1 itself reads and writes / K1 time'},
                         'frac_of_measured_copy': achieved / copy_gbs if copy_gbs else None,
                         'measured_copy_GBps': copy_gbs, 'guide_copy_GBps': HBM_COPY_GUIDE_GBS,
                         'other_kernels_avg_ms': {'finalize': k2_ms / launches, 'combine': k3_ms / launches}},
            'verify': verify,
        }
        if not args.no_cpu and world == 1:           # the CPU baseline is an N=1 figure
            line['cpu_baseline'] = cpu_baseline(cpu_rows[0], cpu_rows[1], n0, n1, method, 7 if args.all_tests else 1, usable_cpus())
    if dist is not None:
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
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
