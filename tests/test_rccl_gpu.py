"""-m gpu: the RCCL ("nccl" backend) reassembly of the per-base tracks on real hardware (SURVEY.md §8e,
BASELINE.json configs[3]).  A 1-GPU box runs the collective branch with one rank (all_gather_into_tensor
through RCCL, device tensors, async work objects); boxes with more GPUs also run one rank per GPU."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _problem(P, n, nb, dev, nm):
    import torch
    L = nm._lib
    det = nm.DeviceDetector(torch.device(dev).index or 0, nb=nb, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    a = torch.empty(P * n, dtype=torch.float32, device=dev); b = torch.empty(P * n, dtype=torch.float32, device=dev)
    det.synth_fill(a, 11, 0, P, 0, n, 700, 0.8); det.synth_fill(b, 11, 0, P, 1, n, 700, 0.8)
    rid = torch.as_tensor((np.arange(P) // 997).astype(np.int32), device=dev)

    def compute(lo, hi):
        r = det.run(a[lo * n:hi * n], b[lo * n:hi * n], rid[lo:hi], stride0=n, stride1=n, npos=hi - lo)
        return {k: v.clone() for k, v in r.items()}
    return det, compute


def _check_rank(rank, world, P, n, nb):
    """body shared by the in-process world-1 test and the one-rank-per-GPU children"""
    import torch
    import torch.distributed as dist
    import nanomod_amd as nm
    from nanomod_amd import sharding
    dev = 'cuda:%d' % rank
    torch.cuda.set_device(rank)
    det, compute = _problem(P, n, nb, dev, nm)
    full = compute(0, P)                                                # unsharded, this device
    tracks = ('ks_p', 'comb_p', 'comb_st')
    got = sharding.sharded_detect(compute, P, nb, tracks=tracks, device=dev, force_collective=True)
    for k in tracks:
        assert got[k].is_cuda and torch.equal(got[k], full[k]), (rank, k)
    for chunks in (1, 3):
        state = sharding.PipelinedGather(P, world, chunks, tracks, dev)
        for _ in range(2):                                              # two steps: buffers reused behind async work
            sharding.pipelined_detect(lambda c, lo, hi: compute(lo, hi), state, nb, force_collective=True)
        assert any(len(w) for w in state.work), 'the collective branch did not run'
        res = state.result()
        for k in tracks:
            assert torch.equal(res[k], full[k]), (rank, chunks, k)
    # more ranks than positions: a rank with an empty shard pads on its device (nccl takes device tensors only)
    tiny = sharding.sharded_detect(compute, min(P, max(world - 1, 1)), nb, tracks=('ks_p',), device=dev, force_collective=True)
    assert tiny['ks_p'].is_cuda and tiny['ks_p'].dtype == torch.float64
    dist.barrier()


@pytest.mark.timeout(300)
def test_rccl_allgather_world1_forced_collective():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group is already initialised in this process')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=torch.device('cuda:0'))
    try:
        assert dist.get_backend() == 'nccl'
        _check_rank(0, 1, 5003, 64, 3)
    finally:
        dist.destroy_process_group()


def _child(rank, world, port, P, n, nb, q):
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda:%d' % rank))
        _check_rank(rank, world, P, n, nb)
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as e:                                                # noqa: BLE001 — reported to the parent
        q.put((rank, repr(e)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world', [2, 8])
def test_rccl_allgather_one_rank_per_gpu(world):
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip('needs %d GPUs on this box' % world)
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')                                         # children start before they touch a GPU
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_child, args=(r, world, port, 20011, 64, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(v == 'ok' for v in results.values()), results


def _cabi_rank(rank, world, id_bytes, P, n, nb):
    """one rank of the torch.distributed-free path: its block through nmod_detect_batch, the tracks through nmod_allgather_tracks"""
    import torch
    import nanomod_amd as nm
    from nanomod_amd import sharding
    dev = 'cuda:%d' % rank
    torch.cuda.set_device(rank)
    det, compute = _problem(P, n, nb, dev, nm)
    full_ref = compute(0, P)
    B = (P + world - 1) // world
    lo, hi = min(rank * B, P), min((rank + 1) * B, P)
    lo_h, hi_h = sharding.halo_bounds(lo, hi, nb, P)
    part = compute(lo_h, hi_h) if hi > lo else None
    tracks = ('ks_p', 'comb_p')
    local = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in tracks]
    if part is not None:
        for t, k in zip(local, tracks):
            t[:hi - lo] = part[k][lo - lo_h: lo - lo_h + hi - lo]
    full = [torch.full((B * world,), -1.0, dtype=torch.float64, device=dev) for _ in tracks]
    comm = sharding.CAbiComm(id_bytes, world, rank, rank)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    for _ in range(2):                                                    # twice: the communicator is reusable, the buffers are rewritten
        comm.allgather(local, full, B, stream=s.cuda_stream)
    s.synchronize()
    comm.close()
    for t, k in zip(full, tracks):
        assert torch.equal(t[:P], full_ref[k]), (rank, k)


@pytest.mark.timeout(300)
def test_cabi_allgather_world1_without_torch_distributed():
    """nmod_comm_* / nmod_allgather_tracks (RCCL bound at run time, no process group): one rank, a side stream, two tracks"""
    import torch.distributed as dist
    from nanomod_amd import sharding
    assert not dist.is_initialized()
    _cabi_rank(0, 1, sharding.CAbiComm.unique_id(), 5003, 64, 3)


def _cabi_child(rank, world, id_bytes, q):
    try:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        _cabi_rank(rank, world, id_bytes, 20011, 64, 2)
        q.put((rank, 'ok'))
    except Exception as e:                                                # noqa: BLE001 — reported to the parent
        q.put((rank, repr(e)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world', [2, 8])
def test_cabi_allgather_one_rank_per_gpu(world):
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip('needs %d GPUs on this box' % world)
    import torch.multiprocessing as mp
    from nanomod_amd import sharding
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    id_bytes = sharding.CAbiComm.unique_id()                              # (made before any child starts; handed over as an argument)
    procs = [ctx.Process(target=_cabi_child, args=(r, world, id_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(v == 'ok' for v in results.values()), results
