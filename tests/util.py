"""Shared helpers for the parity tests: seeded CSR batches that hit every edge of the tile kernel."""
import numpy as np

ALPHABET = np.frombuffer(b"ACGTACGTACGTNnacgtSsRYKMWBDHVU*-.", dtype=np.uint8)

# lengths around the 16-byte vector, the 1 KiB span and the 64 KiB tile of the kernel
EDGE_LENGTHS = [0, 0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 1008, 1023, 1024, 1025, 1040, 2047, 2048, 2049,
                4095, 4096, 4097, 65535, 65536, 65537, 65536 * 2, 65536 * 3 + 5, 200_003]


def random_batch(seed, n_reads, mean_len=600, edge=True, n_channels=512, hours=48, ref=1_528_000_000,
                 invalid_quals=True, fail_frac=0.03, bad_meta=True):
    rng = np.random.default_rng(seed)
    lens = np.rint(rng.lognormal(np.log(mean_len), 1.0, n_reads)).astype(np.int64)
    lens[rng.random(n_reads) < 0.05] = 0
    if edge:
        k = min(len(EDGE_LENGTHS), n_reads)
        lens[rng.choice(n_reads, k, replace=False)] = EDGE_LENGTHS[:k]
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    nb = int(off[-1])
    bases = ALPHABET[rng.integers(0, len(ALPHABET), nb)]
    hi = rng.random(nb) < 0.01          # bytes with the high bit set must never match
    bases = np.where(hi, bases | 0x80, bases).astype(np.uint8)
    bases[rng.random(nb) < 0.002] = 0
    quals = (33 + np.clip(np.rint(rng.normal(14, 7, nb)), 0, 93)).astype(np.uint8)
    if invalid_quals:
        bad = rng.random(nb) < 0.002
        quals[bad] = rng.integers(0, 256, int(bad.sum()), dtype=np.uint8)
    channel = rng.integers(1, n_channels + 1, n_reads).astype(np.uint32)
    mux = rng.integers(1, 5, n_reads).astype(np.uint8)
    t = (ref + rng.integers(0, hours * 3600, n_reads)).astype(np.int64)
    if bad_meta:
        mux[rng.random(n_reads) < 0.02] = 0
        mux[rng.random(n_reads) < 0.01] = 7
        channel[rng.random(n_reads) < 0.01] = 0
        channel[rng.random(n_reads) < 0.01] = n_channels + 3
    ff = (rng.random(n_reads) < fail_frac).astype(np.uint8)
    return dict(bases=bases, quals=quals, offsets=off, channel=channel, mux=mux, time=t, fail_flag=ff)


TABLE_KEYS = ("totals", "len_hist_log2", "len_hist_lin", "q_hist", "gc_hist", "hour_reads", "hour_bases",
              "chan_reads", "chan_bases", "chan_fail", "cmh_reads", "cmh_bases", "chan_min_len", "chan_max_len")
SCALAR_KEYS = ("max_len", "min_pos_len", "lin_bin_width", "n50")


def assert_tables_equal(got, want, ctx=""):
    for k in TABLE_KEYS:
        g, w = np.asarray(got[k]).astype(np.int64), np.asarray(want[k]).astype(np.int64)
        assert g.shape == w.shape, (ctx, k, g.shape, w.shape)
        if not np.array_equal(g, w):
            bad = np.argwhere(g != w)[:5]
            raise AssertionError("%s table %s differs at %s: got %s want %s" % (
                ctx, k, bad.tolist(), [g[tuple(i)] for i in bad], [w[tuple(i)] for i in bad]))
    for k in SCALAR_KEYS:
        assert int(got[k]) == int(want[k]), (ctx, k, got[k], want[k])
