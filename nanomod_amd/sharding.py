"""Position sharding across the GPUs of one node.

Per-position tests are independent and the window combine needs only +-nb
neighbours of the KS p-value track, so every rank takes one contiguous block of
positions and also computes a halo of nb positions on each side instead of
exchanging anything: the data path has no collective.  Each rank ends up with
its own slice of every per-base track (what a per-rank table writer needs);
`gather=True` additionally reassembles the full tracks on every rank with ONE
all-gather per track (RCCL over xGMI when the backend is "nccl").
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


def sharded_detect(compute, npos, nb, tracks=('ks_p', 'comb_p'), group=None, out=None, gather=True):
    """Run `compute(lo_h, hi_h) -> {track: 1-D tensor over [lo_h, hi_h)}` on this rank's block
    (+ halo), drop the halo and — with gather=True — all-gather every requested track.  Returns full-length
    tensors (identical on every rank), or with gather=False this rank's slice [lo, hi) of every track.  `compute` is the HIP path in production; the world_size-2 CPU
    tests inject a checker so the partition / halo / reassembly logic runs under gloo.
    `out`: optional {track: tensor[per * world]} reused across calls (no allocation per step)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(npos, world, rank)
    lo_h, hi_h = halo_bounds(lo, hi, nb, npos)
    local = compute(lo_h, hi_h) if hi > lo else {}
    per = (npos + world - 1) // world
    res = {}
    for name in tracks:
        mine = local[name][lo - lo_h: lo - lo_h + (hi - lo)] if hi > lo else None
        if world == 1 or not gather:
            res[name] = mine
            continue
        if out is not None:
            full = out[name]
            dtype, device = full.dtype, full.device
        else:
            dtype = mine.dtype if mine is not None else torch.float64
            device = mine.device if mine is not None else torch.device('cpu')
            full = torch.empty(per * world, dtype=dtype, device=device)
        if mine is not None and hi - lo == per:
            buf = mine                                   # equal blocks: gather straight from the result
        else:                                            # short or empty last block: pad to the block size
            buf = torch.zeros(per, dtype=dtype, device=device)
            if mine is not None:
                buf[: hi - lo] = mine
        dist.all_gather_into_tensor(full, buf.contiguous(), group=group)
        res[name] = full[:npos]
    return res
