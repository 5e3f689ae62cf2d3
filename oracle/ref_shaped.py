"""Reference-SHAPED CPU baseline (TEST / BENCH INFRASTRUCTURE ONLY, see pypore_oracle.py's header).

What `pypore seqstats` does per read and per channel once the bytes are in memory, organised the way the reference
organises it: a pool of worker processes fed one task per channel (lib/seq_routines.py:916-942); per read ``GC()`` =
six ``count`` passes over the sequence (Bio.SeqUtils.GC, call site lib/fast5_reader.py:58) and a 6-tuple of strings
(:68); per channel the sort / hour wrap / mux aggregation of ``parsing_func`` (:117-158).  HDF5 access,
``dateutil.parse`` and Biopython record construction -- the bulk of the real reference's time -- are NOT included, and
the mean quality is taken as given (the reference reads it from an attribute, :56), so this flatters the CPU side."""
from __future__ import annotations

import multiprocessing as mp
import time

import numpy as np

from . import pypore_oracle as po


_BY_CH = {}     # filled before the pool forks: workers inherit the bytes the way the reference's read them from disk


def _channel_task(args):
    ch_key, ref = args
    reads = _BY_CH[ch_key]
    tuples = []
    for ch, mux, t, seq, q in reads:
        tuples.append(po.get_content_tuple(ch, mux, t, len(seq), q, po.GC(seq)))
    return tuples[0][0], po.parsing_func(tuples, float(ref))


def run(bases, offsets, channel, mux, start_time, mean_q, ref, procs):
    """Returns (seconds, ChannelDict) for the reads of the CSR arrays."""
    by_ch = _BY_CH
    by_ch.clear()
    raw = bases.tobytes()
    for i in range(len(channel)):
        a, e = int(offsets[i]), int(offsets[i + 1])
        by_ch.setdefault(int(channel[i]), []).append((int(channel[i]), int(mux[i]), int(start_time[i]), raw[a:e],
                                                      float(mean_q[i])))
    tasks = [(k, ref) for k in by_ch]
    cd = {}
    t0 = time.perf_counter()
    if procs > 1:
        with mp.get_context("fork").Pool(procs) as pool:
            for ch, res in pool.imap_unordered(_channel_task, tasks, chunksize=4):
                cd[ch] = res
    else:
        for tsk in tasks:
            ch, res = _channel_task(tsk)
            cd[ch] = res
    return time.perf_counter() - t0, cd
