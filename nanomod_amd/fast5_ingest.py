"""FAST5 Events ingest for `detect` -> neutral containers.

Restates the reference's reader — `ReadAllFast5` / `readsubfolder` (myDetect.py:547-633), `mReadSignalBase`
(myDetect.py:33-127) and the HDF5 paths of myFast5.py:92-126 / myCom.py:37-63 — but accumulates flat arrays
instead of dict-of-dict-of-list, and emits the CSR container of `nanomod_amd/container.py` directly.

h5py is imported lazily: this image does not ship it, so the HDF5 access sits behind `reader`, a callable
`path -> (mapped_chrom, mapped_start, mapped_strand, norm_mean[], base[]) | None`; `h5py_reader` is the real
one, tests inject a reader of placeholder files and compare with what the reference's own `ReadAllFast5`
builds from the same reads (tests/golden/fast5_*.npz, made by oracle/gen_golden.py with a stub h5py).
"""
from __future__ import annotations

import os

import numpy as np

EVENTS_PATH = '/Analyses/NanomoCorrected_000/BaseCalled_template/Events'          # myFast5.py:92, myCom.py:48-52
ALIGN_PATH = '/Analyses/NanomoCorrected_000/BaseCalled_template/Alignment'        # myFast5.py:113


def h5py_reader(path):
    """The real reader (needs h5py).  Returns None when the file cannot be opened or has no alignment group,
    as mReadSignalBase does (myDetect.py:41-58)."""
    import h5py
    try:
        f = h5py.File(path, 'r')
    except Exception:
        print('cannot open ' + path)
        return None
    with f:
        if ALIGN_PATH not in f:
            return None
        attrs = dict(f[ALIGN_PATH].attrs.items())                                     # myFast5.py:119-126
        ev = f[EVENTS_PATH][()]
        dec = lambda v: v.decode() if isinstance(v, bytes) else str(v)
        base = np.array([dec(b) for b in ev['base']])
        return dec(attrs['mapped_chrom']), int(attrs['mapped_start']), dec(attrs['mapped_strand']), \
            np.asarray(ev['norm_mean'], dtype=np.float64), base


def read_passes_filters(n_events, mapped_chrom, mapped_start, mapped_strand, opts, log=print, name=''):
    """The per-read filters of mReadSignalBase (myDetect.py:76-103) that do not depend on accumulated state."""
    if 'Chr' in opts and opts['Chr'] != mapped_chrom:
        return False
    if 'Pos2' in opts and (mapped_start > opts['Pos2'] or mapped_start + n_events < opts['Pos']):
        return False
    if 'start_pos' in opts and 'end_pos' in opts:
        if mapped_start > opts['start_pos'] or mapped_start + n_events < opts['end_pos']:
            return False
    min_lr, nbw = opts.get('min_lr', 500), opts.get('min_lr_nb', 0)
    if nbw < 1:
        if n_events < min_lr:                                                          # myDetect.py:90-93
            log('CheckReadMappedLength={} {}'.format(name, n_events))
            return False
    else:
        if not (min_lr - nbw < n_events < min_lr + nbw):
            return False
        end = mapped_start + n_events
        near = lambda v: v < nbw or 8000 - nbw < v < 8000 + nbw or 16000 - nbw < v < 16000 + nbw
        if not (near(mapped_start) and near(end)):                                     # myDetect.py:99-103
            return False
    return True


class GroupBuilder:
    """Accumulates reads of one group; `finish()` returns the container arrays."""

    def __init__(self, opts=None, log=print):
        self.opts = opts or {}
        self.log = log
        self.chrom, self.strand, self.pos, self.val, self.base, self.seq = [], [], [], [], [], []
        self.n_reads = 0

    def add_read(self, mapped_chrom, mapped_start, mapped_strand, norm_mean, base, name=''):
        n = len(norm_mean)
        if not read_passes_filters(n, mapped_chrom, mapped_start, mapped_strand, self.opts, self.log, name):
            return False
        i = np.arange(n, dtype=np.int64)
        pos = i + mapped_start if mapped_strand == '+' else mapped_start + n - 1 - i    # myDetect.py:108-111
        keep = np.ones(n, dtype=bool)
        if 'start_pos' in self.opts and 'end_pos' in self.opts:                         # myDetect.py:112-114
            keep = (pos >= self.opts['start_pos']) & (pos <= self.opts['end_pos'])
        k = int(keep.sum())
        self.chrom.append(np.full(k, mapped_chrom)); self.strand.append(np.full(k, mapped_strand))
        self.pos.append(pos[keep]); self.val.append(np.asarray(norm_mean, dtype=np.float64)[keep])
        self.base.append(np.asarray(base)[keep]); self.seq.append(np.full(k, self.n_reads, dtype=np.int64))
        self.n_reads += 1
        return True

    def finish(self):
        if not self.pos:
            e = np.zeros(0)
            return dict(chrom=np.zeros(0, dtype=str), strand=np.zeros(0, dtype=str), pos=np.zeros(0, np.int64),
                        base=np.zeros(0, dtype=str), off=np.zeros(1, np.int64), sig=e)
        chrom = np.concatenate(self.chrom).astype(str); strand = np.concatenate(self.strand).astype(str)
        pos = np.concatenate(self.pos); val = np.concatenate(self.val)
        base = np.concatenate(self.base).astype(str); seq = np.concatenate(self.seq)
        # group by (chrom, strand, pos); inside a position keep the order the reads were appended in
        order = np.lexsort((seq, pos, strand == '-', chrom))
        chrom, strand, pos, val, base = chrom[order], strand[order], pos[order], val[order], base[order]
        first = np.ones(len(pos), dtype=bool)
        first[1:] = (chrom[1:] != chrom[:-1]) | (strand[1:] != strand[:-1]) | (pos[1:] != pos[:-1])
        starts = np.nonzero(first)[0]
        off = np.append(starts, len(pos)).astype(np.int64)
        last = off[1:] - 1                                  # the base of the LAST read wins (myDetect.py:122)
        return dict(chrom=chrom[starts], strand=strand[starts], pos=pos[starts], base=base[last], off=off, sig=val)


def ingest_folder(folder, opts=None, reader=None, suffix='.fast5', log=print):
    """Breadth-first walk exactly like ReadAllFast5 / readsubfolder (myDetect.py:575-633): files of a folder in
    os.listdir order, sub-folders (except 'mall') queued for the next level."""
    reader = reader or h5py_reader
    gb = GroupBuilder(opts, log)
    level = [folder.rstrip('/')]
    n_files = 0
    while level:
        nxt = []
        for cur in level:
            for name in os.listdir(cur):
                path = cur + '/' + name
                if name.endswith(suffix):
                    n_files += 1
                    if not os.path.isfile(path):
                        continue
                    rec = reader(path)
                    if rec is not None:
                        gb.add_read(*rec, name=path)
                elif os.path.isdir(path) and name != 'mall':
                    nxt.append(path)
        level = nxt
    log('Number of files in ' + str(folder) + 'is ' + str(n_files))
    return gb.finish()
