"""CPU, world_size 2 over gloo: the position partition + halo + all-gather reassembly reproduces the
unsharded tracks bit for bit.  The compute is injected (the C oracle as checker): the production
engine is the HIP path, which needs a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

oracle_c = pytest.importorskip('oracle_c', reason='make -C oracle')


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _make(npos, seed):
    rng = np.random.default_rng(seed)
    n0 = rng.integers(5, 40, npos); n1 = rng.integers(5, 40, npos)
    off0 = np.zeros(npos + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(npos + 1, np.int64); off1[1:] = np.cumsum(n1)
    sig0 = rng.normal(0, 1, off0[-1]).astype(np.float32); sig1 = rng.normal(0.2, 1, off1[-1]).astype(np.float32)
    rid = np.cumsum(rng.random(npos) < 0.05).astype(np.int32)      # runs break independently of the shard cut
    return sig0, off0, sig1, off1, rid


def _worker(rank, world, port, npos, nb, method, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nanomod_amd import sharding
    sig0, off0, sig1, off1, rid = _make(npos, 42)

    def compute(lo, hi):
        o0 = off0[lo:hi + 1] - off0[lo]; o1 = off1[lo:hi + 1] - off1[lo]
        r = oracle_c.detect_batch(sig0[off0[lo]:off0[hi]], o0, sig1[off1[lo]:off1[hi]], o1, rid[lo:hi], nb, 2.0, method, threads=1)
        return {k: torch.from_numpy(v) for k, v in r.items() if k != 'status'}
    out = sharding.sharded_detect(compute, npos, nb, tracks=('ks_p', 'comb_p', 'comb_st'))
    # gather=False (what bench.py times): this rank's slice only, halo dropped, no collective
    mine = sharding.sharded_detect(compute, npos, nb, tracks=('ks_p', 'comb_p', 'comb_st'), gather=False)
    lo, hi = sharding.shard_bounds(npos, world, rank)
    for k, v in mine.items():
        got = np.zeros(0) if v is None else v.numpy()
        assert got.shape[0] == hi - lo and np.array_equal(got, out[k].numpy()[lo:hi], equal_nan=True), (rank, k)
    # block-cyclic pipeline (what bench.py --gpus N times): rounds of `world` blocks, the all-gather of a round issued
    # asynchronously behind its kernels; two steps on one state object (buffer reuse across steps)
    tracks = ('ks_p', 'comb_p', 'comb_st')
    for chunks in (1, 3):
        state = sharding.PipelinedGather(npos, world, chunks, tracks, 'cpu')
        for _ in range(2):
            sharding.pipelined_detect(lambda c, lo, hi: compute(lo, hi), state, nb)
        piped = state.result()
        for k in tracks:
            assert np.array_equal(piped[k].numpy(), out[k].numpy(), equal_nan=True), (rank, chunks, k)
        parts = sharding.pipelined_detect(lambda c, lo, hi: compute(lo, hi), state, nb, gather=False)
        assert len(parts) == chunks
    q.put((rank, {k: v.numpy().copy() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('npos,nb,method', [(501, 2, 'stouffer'), (64, 3, 'fisher'), (3, 2, 'stouffer')])
def test_sharded_equals_unsharded(npos, nb, method):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, npos, nb, method, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sig0, off0, sig1, off1, rid = _make(npos, 42)
    full = oracle_c.detect_batch(sig0, off0, sig1, off1, rid, nb, 2.0, method, threads=1)
    for r in range(world):
        for k in ('ks_p', 'comb_p', 'comb_st'):
            assert np.array_equal(results[r][k], full[k], equal_nan=True), (r, k)


def _pipe_worker(rank, world, port, npos, nb, chunk_list, q):
    """pipelined_detect only (what bench.py --gpus N times), any world size: two consecutive steps per state object"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nanomod_amd import sharding
    sig0, off0, sig1, off1, rid = _make(npos, 7)
    calls = []

    def compute(c, lo, hi):
        calls.append((c, lo, hi))
        o0 = off0[lo:hi + 1] - off0[lo]; o1 = off1[lo:hi + 1] - off1[lo]
        r = oracle_c.detect_batch(sig0[off0[lo]:off0[hi]], o0, sig1[off1[lo]:off1[hi]], o1, rid[lo:hi], nb, 2.0, 'stouffer', threads=1)
        return {k: torch.from_numpy(v) for k, v in r.items() if k != 'status'}
    tracks = ('ks_p', 'comb_p')
    res = {}
    for chunks in chunk_list:
        state = sharding.PipelinedGather(npos, world, chunks, tracks, 'cpu')
        for step in range(2):
            del calls[:]
            sharding.pipelined_detect(compute, state, nb)
            if step == 0:                             # poison the buffers: the second step must rewrite every element
                first = {k: v.numpy().copy() for k, v in state.result().items()}
                for t in tracks:
                    state.full[t].fill_(-7.0)
        second = {k: v.numpy().copy() for k, v in state.result().items()}
        for k in tracks:
            assert np.array_equal(first[k], second[k], equal_nan=True), (rank, chunks, k)
        # this rank computed exactly its non-empty blocks (+ halo), nothing for an empty tail block
        want = []
        for c in range(chunks):
            lo, hi = sharding.cyclic_block(npos, world, rank, chunks, c)
            if hi > lo:
                want.append((c,) + sharding.halo_bounds(lo, hi, nb, npos))
        assert calls == want, (rank, chunks, calls, want)
        res[chunks] = second
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,npos', [(4, 43), (8, 67), (8, 9)])
def test_pipelined_gather_world_4_and_8_short_and_empty_tail_blocks(world, npos):
    """block-cyclic partition at the node sizes the driver scales to: with chunks = 4 the tail holds a short block and
    empty ones (npos = 67, world 8: B = 3, block 22 holds one position, blocks 23..31 none; npos = 9: most ranks idle)"""
    from nanomod_amd import sharding
    chunk_list = (1, 4)
    B = sharding.cyclic_block_len(npos, world, 4)
    sizes = [hi - lo for c in range(4) for r in range(world) for lo, hi in [sharding.cyclic_block(npos, world, r, 4, c)]]
    assert 0 in sizes and any(0 < x < B for x in sizes) or npos == 9
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, npos, 2, chunk_list, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sig0, off0, sig1, off1, rid = _make(npos, 7)
    full = oracle_c.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0, 'stouffer', threads=1)
    for r in range(world):
        for chunks in chunk_list:
            for k in ('ks_p', 'comb_p'):
                assert np.array_equal(results[r][chunks][k], full[k], equal_nan=True), (r, chunks, k)


def test_cyclic_blocks_tile_the_genome():
    from nanomod_amd import sharding
    for npos, world, chunks in ((1000, 2, 4), (1001, 8, 3), (5, 4, 2), (4600000, 8, 4)):
        B = sharding.cyclic_block_len(npos, world, chunks)
        cuts = [sharding.cyclic_block(npos, world, r, chunks, c) for c in range(chunks) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == npos
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(len(cuts) - 1))
        assert all(hi - lo <= B for lo, hi in cuts)


def test_balanced_bounds_cover_and_balance():
    """size-balanced cut for ragged coverage: the blocks tile [0, npos) and carry about equal sample counts"""
    from nanomod_amd import sharding
    rng = np.random.default_rng(3)
    n0 = np.round(rng.lognormal(np.log(1000), 0.5, 5000)).astype(np.int64); n1 = np.round(rng.lognormal(np.log(50), 0.5, 5000)).astype(np.int64)
    off0 = np.r_[0, np.cumsum(n0)]; off1 = np.r_[0, np.cumsum(n1)] + 17
    for world in (1, 2, 8):
        cuts = [sharding.balanced_bounds(off0, off1, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 5000 and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        work = [int((n0 + n1)[lo:hi].sum()) for lo, hi in cuts]
        assert max(work) - min(work) <= 2 * int((n0 + n1).max())
    assert sharding.balanced_bounds(np.zeros(1, np.int64), np.zeros(1, np.int64), 4, 2) == (0, 0)
