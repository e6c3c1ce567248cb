"""Neutral per-group signal container for the `detect` path.

The reference ingests FAST5/HDF5 (myDetect.py:33-127,547-633); h5py is not part of this build, so
the CLI reads one `.npz` per read group holding what ReadAllFast5 accumulates in memory
(`norm_mean[(chrom,strand)][pos] -> samples`, `base[(chrom,strand)][pos]`) as flat arrays:

    chrom  (U)        strand (U1, '+' / '-')    pos (int64, 0-based)    base (U1)
    off    (int64[n+1], CSR row offsets)        sig (float32 | int16 milli-units | float64)
"""
from __future__ import annotations

import numpy as np

FIELDS = ('chrom', 'strand', 'pos', 'base', 'off', 'sig')


def save_group(path, chrom, strand, pos, base, off, sig):
    np.savez_compressed(path, chrom=np.asarray(chrom), strand=np.asarray(strand), pos=np.asarray(pos, dtype=np.int64),
                        base=np.asarray(base), off=np.asarray(off, dtype=np.int64), sig=np.asarray(sig))


def load_group(path):
    z = np.load(path)
    g = {k: z[k] for k in FIELDS}
    n = len(g['pos'])
    if not (len(g['chrom']) == len(g['strand']) == len(g['base']) == n and len(g['off']) == n + 1
            and g['off'][-1] == len(g['sig'])):
        raise ValueError('%s: inconsistent container' % path)
    return g


def from_moptions_dataset(ds):
    """The reference's in-memory dataset (myDetect.py:569-572) -> container arrays (for tests / migration)."""
    chrom, strand, pos, base, chunks = [], [], [], [], []
    for sk in sorted(ds['norm_mean'].keys()):
        for pk in sorted(ds['norm_mean'][sk].keys()):
            chrom.append(sk[0]); strand.append(sk[1]); pos.append(pk); base.append(ds['base'][sk][pk])
            chunks.append(np.asarray(ds['norm_mean'][sk][pk], dtype=np.float64))
    off = np.zeros(len(pos) + 1, dtype=np.int64)
    if pos:
        off[1:] = np.cumsum([len(c) for c in chunks])
    sig = np.concatenate(chunks) if chunks else np.zeros(0)
    return dict(chrom=np.array(chrom, dtype=str), strand=np.array(strand, dtype=str), pos=np.array(pos, dtype=np.int64),
                base=np.array(base, dtype=str), off=off, sig=sig)


def gather_rows(sig, off, rows):
    """CSR sub-selection: the rows `rows` (in that order) as a new (sig, off)."""
    lens = off[rows + 1] - off[rows]
    new_off = np.zeros(len(rows) + 1, dtype=np.int64)
    new_off[1:] = np.cumsum(lens)
    if len(rows) == 0:
        return sig[:0], new_off
    if int(rows[-1]) - int(rows[0]) + 1 == len(rows) and (len(rows) == 1 or bool(np.all(np.diff(rows) == 1))):
        return sig[off[rows[0]]:off[rows[-1] + 1]], new_off          # consecutive rows: a view, no gather
    idx = np.repeat(off[rows] - new_off[:-1], lens) + np.arange(new_off[-1], dtype=np.int64)
    return sig[idx], new_off
