"""Differential fuzz of the tile kernels against the C oracle (GPU box only): many seeded batches whose length
distributions stress read boundaries (runs of 0..40-byte reads, reads ending on vector / span / tile edges, single reads
of several tiles), each through both kernel instantiations and several tile sizes.  Per-read integers must be bit-exact.
usage: python tools/fuzz_parity.py [n_cases] [seed0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pypore_amd import qc
from oracle import c_oracle as co
from util import ALPHABET

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000


def lengths(rng, regime, n):
    if regime == 0:      # tiny reads, many per vector
        return rng.integers(0, 41, n)
    if regime == 1:      # around the vector and the span
        base = rng.choice([15, 16, 17, 31, 32, 33, 1008, 1023, 1024, 1025, 1039, 1040], n)
        return base + rng.integers(-1, 2, n) * (rng.random(n) < 0.3)
    if regime == 2:      # lognormal short
        return np.rint(rng.lognormal(np.log(300), 1.2, n)).astype(np.int64)
    if regime == 3:      # long reads with bursts of tiny ones
        l = np.rint(rng.lognormal(np.log(30000), 1.0, n)).astype(np.int64)
        burst = rng.random(n) < 0.5
        l[burst] = rng.integers(0, 20, int(burst.sum()))
        return l
    if regime == 4:      # multiples of the tile / span so that boundaries sit on the edges
        return rng.choice([1024, 2048, 4096, 65536, 65536 * 2, 16, 32, 0], n)
    l = np.rint(rng.lognormal(np.log(5000), 1.3, n)).astype(np.int64)   # config-2 like with a few giants
    l[rng.random(n) < 0.002] = rng.integers(200_000, 700_000)
    return l


bad = 0
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    regime = case % 6
    n = int(rng.integers(50, 4000)) if regime in (3, 5) else int(rng.integers(200, 30000))
    lens = np.maximum(lengths(rng, regime, n), 0).astype(np.int64)
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    nb = int(off[-1])
    bases = ALPHABET[rng.integers(0, len(ALPHABET), nb)].copy()
    bases[rng.random(nb) < 0.01] |= 0x80
    quals = (33 + np.clip(np.rint(rng.normal(14, 7, nb)), 0, 93)).astype(np.uint8)
    sel = rng.random(nb) < 0.003
    quals[sel] = rng.integers(0, 256, int(sel.sum()), dtype=np.uint8)
    ch = rng.integers(1, 513, n).astype(np.uint32); mux = rng.integers(1, 5, n).astype(np.uint8)
    t = np.full(n, 1_528_000_000, np.int64)
    want = co.per_read(bases, quals, off, nthreads=8)
    werr = [int(e) for e in want["err"]]
    # kernel configurations: what ships (dense_kernel alone for a batch that is dense on average, else the pair with its
    # per-tile probe), the pair at a second threshold, every tile forced sparse (tile_kernel alone), every tile forced
    # onto the dense list, dense_kernel alone over every tile
    for vecb in ("auto", "auto64", "sparse", "dense", "all"):
        for shift in (16, 13, 12):
            tune = dict(tile_shift=shift)
            if vecb == "sparse": tune["tile_dense"] = 0
            if vecb == "dense": tune["tile_dense"] = 1
            if vecb == "all": tune["tile_all"] = 1      # dense_kernel alone over every tile
            if vecb in ("auto64", "sparse", "dense"): tune["tile_all"] = 0   # the pair, whatever the mean
            if vecb == "auto64": tune["dense_len"] = 64
            qc.QCContext.DEFAULT_TUNING = tune
            with qc.QCContext(device=0, ref_time_s=1_528_000_000) as ctx:
                ctx.submit(bases, quals, off, ch, mux, t)
                got = ctx.read_stats()
            ok = all(np.array_equal(got[k], want[k]) for k in ("length", "gc", "n", "qsum"))
            ok = ok and [(int(h) << 32) + int(l) for h, l in zip(got["err_hi"], got["err_lo"])] == werr
            if not ok:
                bad += 1
                k = next(k for k in ("length", "gc", "n", "qsum") if not np.array_equal(got[k], want[k])) if not all(
                    np.array_equal(got[k], want[k]) for k in ("length", "gc", "n", "qsum")) else "err"
                print("MISMATCH case %d regime %d config %s shift %s field %s" % (case, regime, vecb, shift, k), flush=True)
    if case % 10 == 9:
        print("case %d done, %.0f s, %d mismatches" % (case, time.time() - t0, bad), flush=True)
print("fuzz: %d cases x 15 kernel configurations, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
