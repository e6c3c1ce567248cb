"""Position sharding across the GPUs of one node.

Per-position tests are independent and the window combine needs only +-nb
neighbours of the KS p-value track, so a rank computes contiguous blocks of
positions plus a halo of nb recomputed positions on each side of a block:
the arithmetic needs no exchange.  The only collective is the reassembly of
the per-base tracks (BASELINE.json north_star: "RCCL all-gather over xGMI to
reassemble the per-base p-value track"), one all-gather per track and block.

Two partitions:

* `sharded_detect` — one contiguous block per rank (`shard_bounds`, or
  `balanced_bounds` for ragged coverage), one all-gather per track at the end.
* `pipelined_detect` — block-cyclic: the genome is cut into `chunks` rounds of
  `world` equal blocks, rank r owns block r of every round (`cyclic_block`).
  The all-gather of round c then lands as ONE contiguous piece
  [c*world*B, (c+1)*world*B) of the full track in natural position order, and it
  is issued asynchronously (RCCL's own stream) while the kernels of round c+1
  run: of the gather traffic only the last round is exposed.  xGMI is
  point-to-point, so one all-gather already drives all 7 links of a GPU; smaller
  rounds only shorten the exposed tail.

With the "nccl" backend (= RCCL on ROCm) all tensors handed to a collective are
device tensors; ranks with an empty shard allocate their padding on `device`.
"""
from __future__ import annotations


def shard_bounds(npos, world, rank):
    """Contiguous equal blocks of ceil(npos/world) positions; the last block may be short."""
    per = (npos + world - 1) // world
    lo = min(rank * per, npos)
    hi = min(lo + per, npos)
    return lo, hi


def balanced_bounds(off0, off1, world, rank):
    """Contiguous blocks of about equal WORK for ragged coverage (equal sum of n0 + n1, not equal
    position counts): rank r takes the positions whose cumulative sample count falls in its 1/world share.
    off0 / off1: the CSR offsets (host arrays, npos + 1 each)."""
    import numpy as np
    cum = (np.asarray(off0, dtype=np.int64) - off0[0]) + (np.asarray(off1, dtype=np.int64) - off1[0])   # samples before position i
    npos = len(cum) - 1
    total = int(cum[-1])
    cuts = [int(np.searchsorted(cum, total * r // world, side='left')) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, npos
    cuts = [min(max(c, 0), npos) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def halo_bounds(lo, hi, nb, npos):
    return max(lo - nb, 0), min(hi + nb, npos)


def cyclic_block_len(npos, world, chunks):
    """Block length B of the block-cyclic partition: chunks * world blocks of B positions cover [0, npos)."""
    return (npos + world * chunks - 1) // (world * chunks)


def cyclic_block(npos, world, rank, chunks, c):
    """Block of rank `rank` in round `c` of the block-cyclic partition: [lo, hi), possibly short or empty at the end."""
    B = cyclic_block_len(npos, world, chunks)
    lo = min((c * world + rank) * B, npos)
    return lo, min(lo + B, npos)


def _world_rank(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def sharded_detect(compute, npos, nb, tracks=('ks_p', 'comb_p'), group=None, out=None, gather=True, device=None,
                   force_collective=False):
    """Run `compute(lo_h, hi_h) -> {track: 1-D float64 tensor over [lo_h, hi_h)}` on this rank's block
    (+ halo), drop the halo and — with gather=True — all-gather every requested track.  Returns full-length
    tensors (identical on every rank), or with gather=False this rank's slice [lo, hi) of every track.  `compute` is
    the HIP path in production; the world_size-2 CPU tests inject a checker so the partition / halo / reassembly
    logic runs under gloo.
    `out`: optional {track: tensor[per * world]} reused across calls (no allocation per step).
    `device`: where this rank's padding / result buffers live when it has no block of its own (required under the
    nccl backend, whose collectives take device tensors only); tracks are float64 on every rank.
    `force_collective`: issue the all-gather even when world == 1 (exercises the RCCL path on one GPU)."""
    import torch
    import torch.distributed as dist
    world, rank = _world_rank(group)
    lo, hi = shard_bounds(npos, world, rank)
    lo_h, hi_h = halo_bounds(lo, hi, nb, npos)
    local = compute(lo_h, hi_h) if hi > lo else {}
    per = (npos + world - 1) // world
    collective = gather and (world > 1 or (force_collective and dist.is_initialized()))
    res = {}
    for name in tracks:
        mine = local[name][lo - lo_h: lo - lo_h + (hi - lo)] if hi > lo else None
        if not collective:
            res[name] = mine
            continue
        if mine is not None and mine.dtype != torch.float64:
            raise TypeError('track %r must be float64 (every rank gathers the same dtype), got %s' % (name, mine.dtype))
        if out is not None:
            full = out[name]
            dev = full.device
        else:
            dev = mine.device if mine is not None else torch.device(device if device is not None else 'cpu')
            full = torch.empty(per * world, dtype=torch.float64, device=dev)
        if mine is not None and hi - lo == per:
            buf = mine                                   # equal blocks: gather straight from the result
        else:                                            # short or empty last block: pad to the block size
            buf = torch.zeros(per, dtype=torch.float64, device=dev)
            if mine is not None:
                buf[: hi - lo] = mine
        dist.all_gather_into_tensor(full, buf.contiguous(), group=group)
        res[name] = full[:npos]
    return res


class PipelinedGather:
    """State of `pipelined_detect` kept across steps: the full-length track buffers (natural position order, padded to
    chunks * world * B) and the outstanding collectives of the previous step."""

    def __init__(self, npos, world, chunks, tracks, device, dtype=None):
        import torch
        self.npos, self.world, self.chunks, self.tracks = npos, world, chunks, tuple(tracks)
        self.B = cyclic_block_len(npos, world, chunks)
        self.full = {t: torch.empty(self.chunks * self.world * self.B, dtype=dtype or torch.float64, device=device)
                     for t in self.tracks}
        self.pad = None
        self.work = [[] for _ in range(chunks)]

    def wait(self, c=None):
        for cc in (range(self.chunks) if c is None else (c,)):
            for w in self.work[cc]:
                w.wait()
            self.work[cc] = []

    def result(self):
        self.wait()
        return {t: v[:self.npos] for t, v in self.full.items()}


def pipelined_detect(compute, state, nb, group=None, gather=True, force_collective=False):
    """One step of the block-cyclic pipeline.  `compute(c, lo_h, hi_h) -> {track: tensor over [lo_h, hi_h)}` runs the
    hot path on this rank's block of round c (+ halo) and must not reuse the tensors it returned for round c before
    `state.wait(c)` of the next step (pipelined_detect calls it first thing in every round).  With gather=True the
    all-gather of round c is issued with async_op=True right behind the kernels of round c — the collective's stream
    waits for them, the caller's stream does not wait for the collective — and runs beside the kernels of round c+1.
    Returns `state`; `state.result()` waits for the outstanding collectives and returns the full tracks.
    gather=False: compute only (the tracks stay sharded), returns the list of per-round results."""
    import torch
    import torch.distributed as dist
    world, rank = _world_rank(group)
    if world != state.world:
        raise ValueError('PipelinedGather was built for world %d, running with %d' % (state.world, world))
    collective = gather and (world > 1 or (force_collective and dist.is_initialized()))
    B = state.B
    local = []
    for c in range(state.chunks):
        state.wait(c)                                    # round c of the previous step has left its buffers
        lo, hi = cyclic_block(state.npos, world, rank, state.chunks, c)
        lo_h, hi_h = halo_bounds(lo, hi, nb, state.npos)
        res = compute(c, lo_h, hi_h) if hi > lo else {}
        local.append(res)
        if not gather:
            continue
        for name in state.tracks:
            dst = state.full[name][c * world * B:(c + 1) * world * B]
            if hi > lo and (res[name].dtype != dst.dtype or res[name].device != dst.device):
                raise TypeError('track %r: compute returned %s on %s, the gather buffer is %s on %s (every rank gathers the '
                                'same dtype from its own device)' % (name, res[name].dtype, res[name].device, dst.dtype, dst.device))
            if hi - lo == B:
                src = res[name][lo - lo_h: lo - lo_h + B]
            else:                                        # short or empty block at the end of the genome
                if state.pad is None:
                    state.pad = {}
                src = state.pad.setdefault((name, c), torch.zeros(B, dtype=dst.dtype, device=dst.device))
                if hi > lo:
                    src[: hi - lo] = res[name][lo - lo_h: lo - lo_h + (hi - lo)]
            if collective:
                state.work[c].append(dist.all_gather_into_tensor(dst, src.contiguous(), group=group, async_op=True))
            else:                                        # one rank, no process group: the "gather" is a copy
                dst[:B].copy_(src)
    return state if gather else local


# ---------------------------------------------------------------- the same collective without torch.distributed
class CAbiComm:
    """The all-gather of the per-base tracks through the C ABI alone (include/nanomod_hip.h: nmod_comm_*, nmod_allgather_tracks):
    what a caller without torch.distributed — the reference is plain Python / numpy — uses to shard positions over the GPUs of a
    node.  Rank 0 makes the id (`CAbiComm.unique_id()`) and hands its 128 bytes to the others by any means; every rank then
    constructs CAbiComm(id, nranks, rank, device) (collective) and calls allgather() with device pointers.  torch tensors are
    accepted for convenience (their data_ptr() is taken); nothing here needs torch."""

    def __init__(self, unique_id, nranks, rank, device):
        import ctypes as C
        from . import _lib as L
        self._L, self._lib = L, L.load()
        if len(unique_id) != L.COMM_ID_BYTES:
            raise ValueError('unique_id must be %d bytes' % L.COMM_ID_BYTES)
        self._h = C.c_void_p()
        L.check(self._lib.nmod_comm_init_rank(C.c_char_p(bytes(unique_id)), int(nranks), int(rank), int(device), C.byref(self._h)),
                'nmod_comm_init_rank')
        self.nranks, self.rank, self.device = int(nranks), int(rank), int(device)

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib as L
        buf = C.create_string_buffer(L.COMM_ID_BYTES)
        L.check(L.load().nmod_comm_unique_id(buf), 'nmod_comm_unique_id')
        return buf.raw

    def allgather(self, local, full, block_len, stream=0):
        """local[t]: this rank's block (block_len doubles, device memory); full[t]: nranks * block_len doubles on every rank;
        enqueued on `stream` (a hipStream_t value, 0 = the default stream), not synchronised"""
        import ctypes as C
        ptr = lambda x: x.data_ptr() if hasattr(x, 'data_ptr') else int(x)
        n = len(local)
        if len(full) != n:
            raise ValueError('one full track per local track')
        la = (C.c_void_p * n)(*[ptr(x) for x in local]); fa = (C.c_void_p * n)(*[ptr(x) for x in full])
        self._L.check(self._lib.nmod_allgather_tracks(self._h, C.c_void_p(int(stream) or None), int(block_len), n, la, fa), 'nmod_allgather_tracks')

    def close(self):
        if self._h:
            self._L.check(self._lib.nmod_comm_destroy(self._h), 'nmod_comm_destroy')
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
