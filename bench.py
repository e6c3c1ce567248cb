#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (K1 rank statistics + K2 p-values + K3 window
combine) over one rank's shard of the synthetic genome, inputs resident in HBM.
Workload at N=1: BASELINE.json configs[1] — E. coli 4.6 Mb, 200 v 200 reads/position,
KS + weighted Stouffer (window 5), float32 signals.  For N>1 every rank owns 4.6 M
positions of an N x 4.6 M position genome (weak scaling) and computes them with a
+-nb halo of recomputed neighbours: the data path has no collective, every rank keeps its
slice of the per-base tracks in HBM (nanomod_amd/sharding.py; gather=True would reassemble them).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P_ECOLI = 4_600_000
N0 = N1 = 200
NB = 2
WDIF = 2.0
SEED = 20240601
PLANT_PERIOD = 10000
PLANT_SHIFT = 0.8
# s*(n0+n1) + 16 (CSR offsets) + 4 (run id) + 16*k_out + 8 (own-p re-read) = 1660 B
ALGO_BYTES_PER_POS = 4 * (N0 + N1) + 16 + 4 + 16 * 2 + 8
HBM_PEAK_GBS = 8000.0


def cpu_baseline(a, b, threads, target_seconds=15.0, max_positions=1_000_000):
    """The oracle (C restatement, OpenMP over positions) timed on this box's host cores on a bounded
    sample of the same workload: the first positions of this rank's device-resident input, copied
    back.  `a`, `b`: float32 [positions, n] views."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import oracle_c

    def run(npos):
        off0 = np.arange(0, (npos + 1) * N0, N0, dtype=np.int64)
        off1 = np.arange(0, (npos + 1) * N1, N1, dtype=np.int64)
        t0 = time.perf_counter()
        oracle_c.detect_batch(a[:npos].reshape(-1), off0, b[:npos].reshape(-1), off1, np.zeros(npos, np.int32),
                              NB, WDIF, 'stouffer', tests=1, threads=threads)
        return time.perf_counter() - t0
    run(2000)                                    # thread-pool warm-up
    probe = min(20000, a.shape[0])
    rate = probe / run(probe)
    sample = int(min(max_positions, a.shape[0], max(probe, rate * target_seconds)))
    reps = max(1, int(round(rate * target_seconds / sample)))
    dt = sum(run(sample) for _ in range(reps))
    # the reference's own shape of the computation — one scipy-style call sequence per position, one core
    #  — on a small sample of the same rows, for scale: all three tests, as getKStest always does
    import nanomod_oracle as orc
    npy = min(400, a.shape[0])
    t0 = time.perf_counter()
    ksp = [orc.getKStest(a[i].astype(np.float64), b[i].astype(np.float64))[2][1] for i in range(npy)]
    orc.combine_track(np.zeros(npy), np.array(ksp), np.zeros(npy, np.int32), NB, WDIF, orc.METHOD_STOUFFER)
    dpy = time.perf_counter() - t0
    return {'value': sample * reps / dt, 'unit': 'positions/s', 'cores': threads, 'kind': 'port',
            'sample': 'first %d positions of the same workload (200 v 200, KS + Stouffer window 5) x %d passes, '
                      'oracle/nanomod_oracle.c with OpenMP on %d threads (cgroup CPU quota), %.1f s'
                      % (sample, reps, threads, dt),
            'reference_shaped_python': {'value': npy / dpy, 'unit': 'positions/s', 'cores': 1,
                                        'sample': 'oracle/nanomod_oracle.py getKStest + combine per position on the first %d '
                                                  'positions (the reference computes MWU, Welch and KS for every position)' % npy}}


CLOCK_RAMP_STEPS = 8


def usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                         # cgroup v2 CPU quota, if any
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)     # the first ~7 launches run 5-20 % slow (clock ramp, profiles/r1_final_kernel_trace_summary.txt)
    ap.add_argument('--positions', type=int, default=P_ECOLI, help='positions per GPU (default: E. coli 4.6 M)')
    ap.add_argument('--cpu-sample', type=int, default=0, help='cap on positions for the CPU baseline (0 = 1 M)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--all-tests', action='store_true', help='BASELINE.json configs[2]: KS + MWU + Welch-t + Fisher (not the headline metric)')
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device(dev))

    P = args.positions
    total_positions = P * world
    lo, hi = rank * P, (rank + 1) * P
    lo_h, hi_h = sharding.halo_bounds(lo, hi, NB, total_positions)
    n_local = hi_h - lo_h

    det = nm.DeviceDetector(local_rank, nb=NB, weights_dif=WDIF, method='fisher' if args.all_tests else 'stouffer',
                            tests=L.TEST_ALL if args.all_tests else L.TEST_KS)
    sig0 = torch.empty(n_local * N0, dtype=torch.float32, device=dev)
    sig1 = torch.empty(n_local * N1, dtype=torch.float32, device=dev)
    det.synth_fill(sig0, SEED, lo_h, n_local, 0, N0, PLANT_PERIOD, PLANT_SHIFT)
    det.synth_fill(sig1, SEED, lo_h, n_local, 1, N1, PLANT_PERIOD, PLANT_SHIFT)
    rid = torch.zeros(n_local, dtype=torch.int32, device=dev)       # one contiguous run
    out = det.alloc_outputs(n_local)

    def compute(lo_hh, hi_hh):                       # this rank's block + halo is resident: [lo_h, hi_h)
        assert (lo_hh, hi_hh) == (lo_h, hi_h)
        return det.run(sig0, sig1, rid, stride0=N0, stride1=N1, npos=n_local, out=out)

    def step():
        # partition + halo, no collective in the data path (every rank keeps its slice of the tracks in HBM):
        # the code path tests/test_sharding_gloo.py covers
        return sharding.sharded_detect(compute, total_positions, NB, tracks=('ks_p', 'comb_p'), gather=False)

    # The first ~7 kernel launches of a process run 5-20 % slow while the GPU clocks ramp
    # (profiles/r1_final_kernel_trace_summary.txt).  A fixed, untimed ramp precedes the W warm-up steps so that a
    # small W does not put the ramp inside the timed region; it is reported as config.clock_ramp_steps.
    for _ in range(CLOCK_RAMP_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    timer = nm.EventTimer(max(args.steps, 1) + 8)
    det.timer = timer

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    det.timer = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    k1_ms, k1_n = timer.read(L.KERNEL_RANK_STATS)
    k2_ms, _ = timer.read(L.KERNEL_FINALIZE)
    k3_ms, _ = timer.read(L.KERNEL_COMBINE)
    if rank == 0:
        value = total_positions * args.steps / elapsed
        k1_avg_s = (k1_ms / max(k1_n, 1)) * 1e-3
        # measured for the default N=1 workload only (4.6 M positions, KS mode)
        traffic = 7.393e+09 if (P == P_ECOLI and world == 1 and not args.all_tests) else None
        algo_bytes = ALGO_BYTES_PER_POS + (32 if args.all_tests else 0)          # 16 B x 2 more (stat, p) pairs
        achieved = algo_bytes * n_local / k1_avg_s / 1e9 if k1_avg_s > 0 else 0.0
        line = {
            'metric': 'genomic positions/sec (KS + MWU + Welch-t + Fisher)' if args.all_tests else 'genomic positions/sec (KS + Stouffer)', 'value': value, 'unit': 'positions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32 keys / f64 p-values', 'data': 'synthetic',
            'config': {'workload': 'E. coli 4.6 Mb x %d: %d positions/GPU, %d v %d reads/position, KS + weighted '
                                   'Stouffer window=%d (BASELINE.json configs[1])' % (world, P, N0, N1, 2 * NB + 1),
                       'positions_per_gpu': P, 'n0': N0, 'n1': N1, 'clock_ramp_steps': CLOCK_RAMP_STEPS, 'neighborPvalues': NB, 'WeightsDif': WDIF,
                       'parallelism': 'position-sharded x%d, +-%d halo recomputed, no data-path collective' % (world, NB)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_note': 'HBM bytes per launch from rocprofv3 PMC passes of this command (FETCH_SIZE x2 per the '
                                         'gfx950 correction + WRITE_SIZE), profiles/r1_final_pmc_summary.txt; not re-measured in this run',
                         'kernel': 'rank_all_kernel<16,16,f32>' if args.all_tests else 'ks_rank_kernel<16,16,f32>', 'kernel_avg_ms': k1_avg_s * 1e3,
                         'algorithmic_bytes_per_position': algo_bytes,
                         'other_kernels_avg_ms': {'finalize': k2_ms / max(k1_n, 1), 'combine': k3_ms / max(k1_n, 1)}},
        }
        if not args.no_cpu and not args.all_tests and world == 1:   # the CPU baseline is an N=1 figure
            threads = usable_cpus()
            cap = args.cpu_sample or 1_000_000
            cap = min(cap, n_local)
            a = sig0[:cap * N0].cpu().numpy().reshape(cap, N0)
            b = sig1[:cap * N1].cpu().numpy().reshape(cap, N1)
            line['cpu_baseline'] = cpu_baseline(a, b, threads, max_positions=cap)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
