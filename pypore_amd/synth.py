"""Synthetic workloads of BASELINE.md, host side.

``stream_bytes`` is the numpy twin of csrc/ppqc_synth.hip: byte i of each stream depends only on
(seed, i), so the full-size buffers are generated in HBM by the kernel and any slice can be
regenerated here for parity checks.  Read lengths and metadata are small and come from numpy's
PCG64 (deterministic for a given numpy version).
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xD1342543DE82EF95)

CONFIGS = {
    # name: (reads, mu, sigma, n_channels, n_time_bins, hours)
    "config2": dict(n_reads=1_000_000, mu=8.5172, sigma=1.1774, n_channels=512, n_time_bins=48, hours=48),
    "config3": dict(n_reads=10_000_000, mu=7.8240, sigma=1.1774, n_channels=3000, n_time_bins=72, hours=72),
    "config4": dict(n_reads=200_000, mu=10.1266, sigma=1.1774, n_channels=512, n_time_bins=48, hours=48),
}


def _mix64(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def stream_bytes(first, n, seed):
    """(bases, quals) uint8 arrays for stream bytes [first, first + n)."""
    first, n = int(first), int(n)
    if n == 0:
        return np.zeros(0, np.uint8), np.zeros(0, np.uint8)
    w0, w1 = first // 4, (first + n + 3) // 4
    j = np.arange(w0, w1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        hb = _mix64(np.uint64(seed) ^ ((np.uint64(2) * j) * _M))
        hq = _mix64(np.uint64(seed) ^ ((np.uint64(2) * j + np.uint64(1)) * _M))
    bases = np.empty((len(j), 4), np.uint8)
    quals = np.empty((len(j), 4), np.uint8)
    for b in range(4):
        v = ((hb >> np.uint64(16 * b)) & np.uint64(0xFFFF)).astype(np.uint32)
        odd = (v & 1).astype(bool)
        c = np.where(odd, ord("A"), ord("T"))
        c = np.where(v < 26874, np.where(odd, ord("G"), ord("C")), c)
        c = np.where(v < 7, ord("N"), c)
        bases[:, b] = c
        v = ((hq >> np.uint64(16 * b)) & np.uint64(0xFFFF)).astype(np.int64)
        s = (v & 15) + ((v >> 4) & 15) + ((v >> 8) & 15) + (v >> 12)
        q = np.clip((4 * s) // 9 - 1, 1, 60)
        quals[:, b] = 33 + q
    lo = first - 4 * w0
    return bases.reshape(-1)[lo:lo + n].copy(), quals.reshape(-1)[lo:lo + n].copy()


def read_lengths(n_reads, mu, sigma, seed, max_len=2_000_000):
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.clip(np.rint(rng.lognormal(mu, sigma, n_reads)), 1, max_len).astype(np.uint32)


def metadata(n_reads, n_channels, hours, seed, ref_time_s=1_528_000_000):
    """channel uniform 1..n_channels, mux uniform 1..4 with 1 % mux 0, start times uniform over `hours`."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    channel = rng.integers(1, n_channels + 1, n_reads, dtype=np.uint32)
    mux = rng.integers(1, 5, n_reads, dtype=np.uint8)
    mux[rng.random(n_reads) < 0.01] = 0
    t = (ref_time_s + rng.integers(0, hours * 3600, n_reads)).astype(np.int64)
    return channel, mux, t


def adversarial_lengths(n_reads, seed, n_long=100, long_len=2_000_000, short_len=200):
    """Config-4 stress: 99+ % reads of 200 b plus `n_long` reads of 2 Mb scattered through the batch."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.full(n_reads, short_len, dtype=np.uint32)
    lens[rng.choice(n_reads, size=min(n_long, n_reads), replace=False)] = long_len
    return lens


def offsets_from_lengths(lengths):
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    np.cumsum(lengths, dtype=np.uint64, out=off[1:])
    return off
