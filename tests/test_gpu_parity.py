"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Integers, histograms, tables and N50 must be bit-exact; mean quality within 1e-6 relative
(north_star), and within 1e-12 of the fixed-point model it is computed from."""
import numpy as np
import pytest

from util import random_batch, assert_tables_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qc():
    from pypore_amd import qc as m
    return m


def _check_reads(got, want, flagged=None):
    n = len(want["length"])
    fl = np.zeros(n, bool) if flagged is None else np.asarray(flagged).astype(bool)
    keep = ~fl
    for k in ("length", "gc", "n", "qsum"):
        g, w = got[k][keep], want[k][keep]
        if not np.array_equal(g, w):
            i = np.nonzero(g != w)[0][:5]
            raise AssertionError("per-read %s differs at %s: got %s want %s (len %s)" % (
                k, np.nonzero(keep)[0][i], g[i], w[i], want["length"][keep][i]))
    for k in ("length", "gc", "n", "qsum", "err_hi", "err_lo"):
        assert not got[k][fl].any(), k
    assert np.all(got["mean_q"][fl] == 6.99)
    gerr = [(int(h) << 32) + int(l) for h, l in zip(got["err_hi"], got["err_lo"])]
    idx = np.nonzero(keep)[0]
    bad = [i for i in idx if gerr[i] != int(want["err"][i])]
    assert not bad, ("fixed-point error sums differ", bad[:5], [gerr[i] for i in bad[:5]],
                     [int(want["err"][i]) for i in bad[:5]])
    assert np.all(got["err_lo"] < (1 << 32))
    np.testing.assert_allclose(got["mean_q"][keep], want["mean_q"][keep], rtol=1e-12, atol=1e-12)


def _run(qc, b, cfg, c_oracle, use_ff=True, split=None):
    ff = b["fail_flag"] if use_ff else None
    with qc.QCContext(device=0, **cfg) as ctx:
        if split is None:
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], ff)
        else:   # stream the same reads as several host batches through the two staging slots
            R = len(b["channel"])
            cuts = np.linspace(0, R, split + 1).astype(int)
            for k in range(split):
                a, e = cuts[k], cuts[k + 1]
                o = b["offsets"][a:e + 1]
                s0, s1 = int(o[0]), int(o[-1])
                ctx.submit(b["bases"][s0:s1], b["quals"][s0:s1], o - o[0], b["channel"][a:e], b["mux"][a:e],
                           b["time"][a:e], None if ff is None else ff[a:e], slot=k % 2)
        ctx.sync()
        got = ctx.read_stats()
        tables = ctx.finalize()
        hours = ctx.read_hours()
    want = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=4)
    _check_reads(got, want, ff)
    wt = c_oracle.tables(want, b["channel"], b["mux"], b["time"], ff, cfg)
    assert_tables_equal(tables, wt, str(cfg))
    return got, tables, hours


BASE_CFG = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0, tz_offset_s=0,
                ref_time_s=1_528_000_000, fail_q=7.0)


@pytest.mark.parametrize("seed,n_reads,mean_len", [(1, 300, 300), (2, 4000, 900), (3, 20000, 3000), (4, 64, 40000)])
def test_per_read_and_tables_mode0(qc, c_oracle, seed, n_reads, mean_len):
    b = random_batch(seed, n_reads, mean_len)
    _run(qc, b, BASE_CFG, c_oracle)


@pytest.mark.parametrize("seed,n_reads,n_channels", [(11, 500, 12), (12, 6000, 512), (13, 3000, 3)])
def test_tables_mode1_hour_wrap_quirk(qc, c_oracle, seed, n_reads, n_channels):
    cfg = dict(BASE_CFG, time_mode=1, n_channels=n_channels, tz_offset_s=3600)
    b = random_batch(seed, n_reads, 200, n_channels=n_channels, hours=47)
    _run(qc, b, cfg, c_oracle)


def test_promethion_shape_3000_channels_72_bins(qc, c_oracle):
    cfg = dict(BASE_CFG, n_channels=3000, n_time_bins=72)
    b = random_batch(21, 30000, 150, n_channels=3000, hours=75)   # some reads beyond the last bin
    _, tables, _ = _run(qc, b, cfg, c_oracle)
    assert tables["totals"][8] > 0   # time_oob exercised


def test_streamed_batches_equal_single_batch(qc, c_oracle):
    b = random_batch(31, 5000, 700)
    g1, t1, h1 = _run(qc, b, BASE_CFG, c_oracle)
    g2, t2, h2 = _run(qc, b, BASE_CFG, c_oracle, split=7)
    for k in g1:
        assert np.array_equal(g1[k], g2[k]), k
    assert_tables_equal(t1, t2)
    assert np.array_equal(h1, h2)


def test_batch_time_extremes_equal_finalize_time_extremes(qc, c_oracle):
    """A batch leaves its index-only results (extremes, hours, sort keys) when it is submitted, beside the tile kernel;
    the same tables must come out when that is not possible: reference time set after the batches (hours stale),
    a second finalize of the same run, batches added behind a finalize, and a reference time that moves reads out of
    the binned range.  All against the oracle."""
    b = random_batch(77, 6000, 800)
    b["channel"][::97] = 0            # invalid channels: no sort key, no extremes, hour -1
    b["fail_flag"][::13] = 1          # flagged reads count with length 0
    ff = b["fail_flag"]
    want = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=4)

    def submit(ctx, a, e, slot=0):
        o = b["offsets"][a:e + 1]
        s0, s1 = int(o[0]), int(o[-1])
        ctx.submit(b["bases"][s0:s1], b["quals"][s0:s1], o - o[0], b["channel"][a:e], b["mux"][a:e], b["time"][a:e],
                   ff[a:e], slot=slot)

    R = len(b["channel"])
    for ref in (BASE_CFG["ref_time_s"], BASE_CFG["ref_time_s"] + 5 * 3600):
        cfg = dict(BASE_CFG, ref_time_s=ref)
        wt = c_oracle.tables(want, b["channel"], b["mux"], b["time"], ff, cfg)
        # (a) the early path as it is
        with qc.QCContext(device=0, **cfg) as ctx:
            submit(ctx, 0, R)
            t_pre = ctx.finalize()
            h_pre = ctx.read_hours()
            assert_tables_equal(t_pre, wt, "early")
            # (b) a second finalize of the same run goes through the finalize-time kernels
            assert_tables_equal(ctx.finalize(), wt, "second finalize")
            assert np.array_equal(ctx.read_hours(), h_pre)
        # (c) the reference time arrives after the batches
        with qc.QCContext(device=0, **dict(cfg, ref_time_s=ref - 7 * 3600)) as ctx:
            submit(ctx, 0, R // 2, 0)
            submit(ctx, R // 2, R, 1)
            ctx.set_ref_time(ref)
            assert_tables_equal(ctx.finalize(), wt, "late reference time")
            assert np.array_equal(ctx.read_hours(), h_pre)
        # (d) batches behind a finalize, and a fresh run on the same context behind ppqc_reset
        with qc.QCContext(device=0, **cfg) as ctx:
            submit(ctx, 0, R // 3, 0)
            ctx.finalize()
            submit(ctx, R // 3, R, 1)
            assert_tables_equal(ctx.finalize(), wt, "batches behind a finalize")
            ctx.reset()
            submit(ctx, 0, R // 2, 0)
            submit(ctx, R // 2, R, 1)
            assert_tables_equal(ctx.finalize(copy=False), wt, "behind reset")


@pytest.mark.parametrize("seed,mode", [(31, 1), (32, 1), (33, 0)])
def test_single_read_channel_compat_and_missing_start_times(qc, c_oracle, seed, mode):
    """PPQC_COMPAT_MULTI_READ_SINGLE (channel_parser's `if len(l) > 1`, lib/multi_read_fast5_reader.py:64-67: the only
    read of a channel is taken under every mux) against the oracle's table layer, which tests/test_oracle.py holds to
    parsing_func(multi_read_single_quirk=True); and reads without a start time (INT64_MIN)."""
    cfg = dict(BASE_CFG, time_mode=mode, n_channels=64, compat_flags=1)
    b = random_batch(seed, 90, 200, n_channels=64, hours=47)      # most channels hold one read or none
    assert (np.bincount(b["channel"], minlength=70)[1:65] == 1).sum() >= 10
    b["time"] = b["time"].copy()
    b["time"][::11] = -(1 << 63)
    got, tables, hours = _run(qc, b, cfg, c_oracle)
    assert int(tables["totals"][11]) == len(b["time"][::11][(b["channel"][::11] >= 1) & (b["channel"][::11] <= 64)])
    plain = dict(cfg, compat_flags=0)
    want = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=2)
    wt = c_oracle.tables(want, b["channel"], b["mux"], b["time"], b["fail_flag"], plain)
    assert not np.array_equal(np.asarray(tables["cmh_reads"]), np.asarray(wt["cmh_reads"]))    # the bit does something
    assert np.array_equal(np.asarray(tables["chan_reads"]), np.asarray(wt["chan_reads"]))


@pytest.mark.parametrize("n_channels,n_mux,n_time_bins", [(3000, 1, 2048), (8190, 4, 2), (8192, 2, 3), (9000, 1, 48)])
def test_geometries_at_the_lds_limits(qc, c_oracle, n_channels, n_mux, n_time_bins):
    """Configurations whose per-channel accumulators do not fit the kernels' LDS layouts (3000 channels x 2048 bins is
    73 KB in tables_kernel; 8189..8192 channels overflow extremes_kernel's 64 KiB with its static words): they take the
    global-atomic paths instead of failing at finalize time."""
    cfg = dict(BASE_CFG, n_channels=n_channels, n_mux=n_mux, n_time_bins=n_time_bins, bin_seconds=60)
    b = random_batch(n_channels, 3000, 150, n_channels=n_channels, hours=30)
    _run(qc, b, cfg, c_oracle)


def test_no_fail_flag_and_all_flagged(qc, c_oracle):
    b = random_batch(41, 800, 500)
    _run(qc, b, BASE_CFG, c_oracle, use_ff=False)
    b["fail_flag"][:] = 1
    _run(qc, b, BASE_CFG, c_oracle)


def test_empty_and_degenerate_batches(qc, c_oracle):
    z8, z64 = np.zeros(0, np.uint8), np.zeros(1, np.uint64)
    with qc.QCContext(device=0, **BASE_CFG) as ctx:
        ctx.submit(z8, z8, z64, np.zeros(0, np.uint32), z8, np.zeros(0, np.int64))
        t = ctx.finalize()
        assert t["totals"].sum() == 0 and t["n50"] == 0 and t["max_len"] == 0
    # only empty reads
    b = dict(bases=z8, quals=z8, offsets=np.zeros(6, np.uint64), channel=np.arange(1, 6, dtype=np.uint32),
             mux=np.ones(5, np.uint8), time=np.full(5, 1_528_000_000, np.int64), fail_flag=np.zeros(5, np.uint8))
    _run(qc, b, BASE_CFG, c_oracle)
    # a single one-base read, a single read of exactly one tile, and one just over a tile
    for L in (1, 65536, 65537):
        rng = np.random.default_rng(L)
        b = dict(bases=rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L), quals=rng.integers(33, 80, L).astype(np.uint8),
                 offsets=np.array([0, L], np.uint64), channel=np.array([7], np.uint32), mux=np.array([2], np.uint8),
                 time=np.array([1_528_000_100], np.int64), fail_flag=np.zeros(1, np.uint8))
        _run(qc, b, BASE_CFG, c_oracle)


def test_known_answers(qc):
    """Hand-derived KATs of SURVEY.md 8a-2 / 8a-11."""
    recs = [(b"ACTGN", b"!+5?I"), (b"GGCCSsgcAT", b"IIIIIIIIII"), (b"", b""), (b"nNnN", b"!!!!")]
    bases, quals, off = qc.pack_reads(recs)
    with qc.QCContext(device=0, **BASE_CFG) as ctx:
        ctx.submit(bases, quals, off, np.ones(4, np.uint32), np.ones(4, np.uint8), np.full(4, 1_528_000_000, np.int64))
        s = ctx.read_stats()
    assert s["length"].tolist() == [5, 10, 0, 4]
    assert s["gc"].tolist() == [2, 8, 0, 0]          # GC("ACTGN") = 40.0, GC("GGCCSsgcAT") = 80.0
    assert s["n"].tolist() == [1, 0, 0, 4]
    assert s["qsum"].tolist() == [100, 400, 0, 0]
    assert abs(s["mean_q"][0] - 6.5321685674187755) < 1e-6 * 6.53
    assert abs(s["mean_q"][1] - 40.0) < 1e-6 * 40 and s["mean_q"][2] == 0.0 and abs(s["mean_q"][3]) < 1e-9
    assert s["flags"].tolist() == [1, 0, 1, 1]       # 6.53 < 7 fail; 40 pass; empty read q=0 fail; q=0 fail


def test_mean_q_matches_float_spec(qc):
    """north_star: mean quality within 1e-6 relative of -10 log10(mean 10^(-q/10)) in plain fp64."""
    from oracle import pypore_oracle as po
    b = random_batch(51, 400, 800, invalid_quals=False)
    with qc.QCContext(device=0, **BASE_CFG) as ctx:
        ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"])
        s = ctx.read_stats()
    for i in range(len(b["channel"])):
        a, e = int(b["offsets"][i]), int(b["offsets"][i + 1])
        spec = po.mean_q_ont(b["quals"][a:e].tobytes())
        assert abs(s["mean_q"][i] - spec) <= 1e-6 * max(abs(spec), 1e-30), (i, s["mean_q"][i], spec)


def test_radix_sort_and_n50(qc, c_oracle):
    import torch
    rng = np.random.default_rng(7)
    with qc.QCContext(device=0, **BASE_CFG) as ctx:
        # 1.5 M then 2^21 in one context: the second count fits the capacity of the first growth but needs MORE waves than
        # the capacity itself would (the wave count is not monotone in the key count; the counters are sized by a bound)
        for n in (1, 2, 63, 64, 65, 2047, 2048, 2049, 100_000, 1_500_000, 1 << 21, 3_000_001):
            keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
            if n > 1000:
                keys[: n // 3] &= 0xFFFF     # many equal high digits
                keys[n // 3: n // 2] = 12345  # long runs of equal keys
            t = torch.from_numpy(keys.view(np.int32).copy()).cuda()
            ctx.sort_u32_device(t.data_ptr(), n)
            ctx.sync()
            got = t.cpu().numpy().view(np.uint32)
            assert np.array_equal(got, np.sort(keys)), n
    for lens in ([2, 2, 2, 3, 3, 4, 8, 8], [10, 20, 30, 40, 50], [5], [0, 0, 0], [7] * 5000,
                 rng.integers(1, 200_000, 7_001).tolist(),
                 [17_000_000, 5, 300, 70_000, 16_777_216, 1],     # keys with a fourth byte: all four gated passes run
                 [255, 3, 200], [256, 3, 200], [65_535, 65_536, 9]):   # one, two and three passes
        R = len(lens)
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
        nb = int(off[-1])
        with qc.QCContext(device=0, **BASE_CFG) as ctx:
            ctx.submit(np.full(nb, ord("A"), np.uint8), np.full(nb, 50, np.uint8), off, np.ones(R, np.uint32),
                       np.ones(R, np.uint8), np.full(R, 1_528_000_000, np.int64))
            t = ctx.finalize()
        assert t["n50"] == c_oracle.n50(lens), (lens[:8], t["n50"])
    assert c_oracle.n50([2, 2, 2, 3, 3, 4, 8, 8]) == 8 and c_oracle.n50([10, 20, 30, 40, 50]) == 40
    # the N50 kernel's block structure (2048 keys a block): one block, block edges, the N50 in the top block, in a
    # middle block, behind whole blocks of zeros; twice per context: the ticket counter must be back at zero
    for n, kind in [(1, "flat"), (2047, "flat"), (2048, "flat"), (2049, "flat"), (4096, "top"), (4097, "top"),
                    (300_001, "flat"), (300_001, "top"), (300_001, "zeros"), (2048 * 257 + 5, "flat")]:
        if kind == "flat":
            lens = np.full(n, 1000, np.uint32) + (np.arange(n) % 7).astype(np.uint32)
        elif kind == "top":                                   # one key carries more than half the total
            lens = np.full(n, 3, np.uint32)
            lens[n // 2] = 3 * n + 10
        else:                                                 # mostly zeros: the sorted array begins with empty blocks
            lens = np.zeros(n, np.uint32)
            lens[::1000] = 5000 + np.arange(len(lens[::1000]), dtype=np.uint32)
        with qc.QCContext(device=0, **BASE_CFG) as ctx:
            for _ in range(2):
                ctx.reset()
                ctx.submit_records(lens, np.full(n, 9.0), np.ones(n, np.uint32), np.ones(n, np.uint8),
                                   np.full(n, 1_528_000_000, np.int64))
                assert ctx.finalize()["n50"] == c_oracle.n50(lens), (n, kind)
    # lengths from a table (no bytes behind them): the sort gate is max_len from the extremes kernel, up to 2^32 - 2
    for lens in ([4_294_967_294, 7, 4_000_000_000, 123_456_789], [16_777_215, 16_777_216, 3], [1]):
        R = len(lens)
        with qc.QCContext(device=0, **BASE_CFG) as ctx:
            ctx.submit_records(np.asarray(lens, np.uint32), np.full(R, 9.0), np.ones(R, np.uint32), np.ones(R, np.uint8),
                               np.full(R, 1_528_000_000, np.int64))
            t = ctx.finalize()
        assert t["n50"] == c_oracle.n50(lens) and t["max_len"] == max(lens), (lens, t["n50"])


def test_synth_generator_matches_numpy(qc):
    import torch
    from pypore_amd import synth
    with qc.QCContext(device=0, **BASE_CFG) as ctx:
        for first, n in ((0, 4096), (5, 100_003), (1 << 33, 70_001), (3, 1)):
            tb = torch.zeros(n + 32, dtype=torch.uint8, device="cuda")
            tq = torch.zeros(n + 32, dtype=torch.uint8, device="cuda")
            ctx.synth_fill_device(tb.data_ptr(), tq.data_ptr(), first, n, 20240601)
            ctx.sync()
            hb, hq = synth.stream_bytes(first, n, 20240601)
            assert np.array_equal(tb.cpu().numpy()[:n], hb) and np.array_equal(tq.cpu().numpy()[:n], hq)
            assert not tb.cpu().numpy()[n:].any()


def test_device_batch_config4_adversarial_and_determinism(qc, c_oracle):
    """Ultra-long reads split over many wavefronts + thousands of tiny reads, generated in HBM; checked against
    the oracle on the regenerated host copy, and bit-identical across two runs."""
    import torch
    from pypore_amd import synth
    seed = 99
    lens = synth.adversarial_lengths(20_000, seed, n_long=6, long_len=2_000_000)
    lens[:50] = synth.read_lengths(50, 10.1266, 1.1774, seed)
    off = synth.offsets_from_lengths(lens)
    nb, R = int(off[-1]), len(lens)
    ch, mux, t = synth.metadata(R, 512, 48, seed)
    outs = []
    with qc.QCContext(device=0, **BASE_CFG) as ctx:
        tb = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
        tq = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
        ctx.synth_fill_device(tb.data_ptr(), tq.data_ptr(), 0, nb, seed)
        d = [torch.from_numpy(x).cuda() for x in (off.view(np.int64), ch.view(np.int32), mux, t)]
        for _ in range(2):
            ctx.reset()
            ctx.submit_device(R, nb, tb.data_ptr(), tq.data_ptr(), *[x.data_ptr() for x in d])
            outs.append((ctx.read_stats(), ctx.finalize()))
    hb, hq = synth.stream_bytes(0, nb, seed)
    want = c_oracle.per_read(hb, hq, off, nthreads=8)
    _check_reads(outs[0][0], want)
    assert_tables_equal(outs[0][1], c_oracle.tables(want, ch, mux, t, None, BASE_CFG))
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    assert_tables_equal(outs[0][1], outs[1][1])


def _record_columns(seed, n, n_channels, hours=47, tick=1):
    rng = np.random.default_rng(seed)
    length = rng.integers(0, 60000, n).astype(np.uint32)
    mean_q = np.round(rng.normal(9.5, 2.5, n), 3)
    mean_q[:8] = (7.0, 6.999, 6.0, 5.999, 13.999, 14.0, 0.0, 93.0)     # both sides of every threshold
    gc = (length * rng.uniform(0.2, 0.7, n)).astype(np.uint32)
    gc[rng.random(n) < 0.1] = 0
    channel = rng.integers(1, n_channels + 1, n).astype(np.uint32)
    channel[rng.random(n) < 0.01] = n_channels + 5
    mux = np.ones(n, np.uint8)
    t = (1_528_000_000 * tick + rng.integers(0, hours * 3600 * tick, n)).astype(np.int64)
    return length, mean_q, gc, channel, mux, t


def _record_stats(length, mean_q, gc):
    n = len(length)
    return dict(length=length, gc=gc, n=np.zeros(n, np.uint32), qsum=np.zeros(n, np.int64), err=[0] * n,
                mean_q=mean_q, is_record=np.ones(n, np.uint8))


@pytest.mark.parametrize("mode", [0, 1])
def test_records_path_sequencing_summary_rows(qc, c_oracle, mode):
    """ppqc_submit_records: rows of sequencing_summary.txt (lib/seq_routines.py:831-863), mux-less tables, the float
    fail rule of :855."""
    from oracle import pypore_oracle as po
    cfg = dict(BASE_CFG, n_mux=1, time_mode=mode, n_channels=64, tz_offset_s=7200)
    length, mean_q, gc, channel, mux, t = _record_columns(61 + mode, 4000, 64)
    with qc.QCContext(device=0, **cfg) as ctx:
        for a, e in ((0, 1500), (1500, 1500), (1500, 4000)):      # appended in pieces, one of them empty
            ctx.submit_records(length[a:e], mean_q[a:e], channel[a:e], mux[a:e], t[a:e], gc[a:e])
        got = ctx.read_stats()
        tables = ctx.finalize()
    assert np.array_equal(got["length"], length) and np.array_equal(got["gc"], gc)
    assert np.array_equal(got["mean_q"], mean_q)
    assert np.array_equal(got["flags"], (4 | (mean_q < 7.0)).astype(np.uint8))
    stats = _record_stats(length, mean_q, gc)
    wt = c_oracle.tables(stats, channel, mux, t, None, cfg)
    assert_tables_equal(tables, wt, "records mode %d" % mode)
    small = slice(0, 600)
    with qc.QCContext(device=0, **cfg) as ctx:
        ctx.submit_records(length[small], mean_q[small], channel[small], mux[small], t[small], gc[small])
        ts = ctx.finalize()
    assert_tables_equal(ts, po.build_tables(_record_stats(length[small], mean_q[small], gc[small]), channel[small],
                                            mux[small], t[small], None, cfg), "records vs python oracle")


def test_cell_tables_on_both_sides_of_the_packing_bound(qc, c_oracle):
    """tables_kernel packs a cell's read count and bases into one atomic when n < 2^22 and n * max_len < 2^42 (decided on
    the device); otherwise two atomics.  Lengths from a table reach both sides with few reads: the cell tables (and
    everything else) must equal the oracle's either way, also with every read in ONE cell."""
    cfg = dict(BASE_CFG, n_mux=1, time_mode=0, n_channels=16)
    n = 3000
    for top in (60_000, (1 << 42) // n - 1, (1 << 42) // n + 2, 4_294_967_294):
        length, mean_q, gc, channel, mux, t = _record_columns(91, n, 16)
        rng = np.random.default_rng(int(top) % 9973)
        length = rng.integers(0, min(top, 4_294_967_294) + 1, n).astype(np.uint64)
        length[0] = top                                        # max_len is exactly `top`
        length = length.astype(np.uint32)
        gc = np.zeros(n, np.uint32)
        for one_cell in (False, True):
            ch = np.full(n, 3, np.uint32) if one_cell else channel
            tt = np.full(n, 1_528_000_000 + 5 * 3600, np.int64) if one_cell else t
            with qc.QCContext(device=0, **cfg) as ctx:
                ctx.submit_records(length, mean_q, ch, mux, tt, gc)
                tables = ctx.finalize()
            wt = c_oracle.tables(_record_stats(length, mean_q, gc), ch, mux, tt, None, cfg)
            assert_tables_equal(tables, wt, "top %d one_cell %s" % (top, one_cell))
            if one_cell:
                assert int(tables["cmh_reads"][2, 0, 5]) == n and int(tables["cmh_bases"][2, 0, 5]) == int(length.astype(np.uint64).sum())


def test_records_and_csr_batches_mix_in_one_run(qc, c_oracle):
    b = random_batch(71, 1200, 400)
    length, mean_q, gc, channel, mux, t = _record_columns(72, 900, 512, hours=48)
    with qc.QCContext(device=0, **BASE_CFG) as ctx:
        ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
        ctx.submit_records(length, mean_q, channel, mux, t, gc)
        tables = ctx.finalize()
    w = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=2)
    nb = len(b["channel"])
    stats = dict(length=np.concatenate([w["length"], length]), gc=np.concatenate([w["gc"], gc]),
                 n=np.concatenate([w["n"], np.zeros(len(length), np.uint32)]),
                 qsum=np.concatenate([w["qsum"], np.zeros(len(length), np.int64)]),
                 err=list(w["err"]) + [0] * len(length), mean_q=np.concatenate([w["mean_q"], mean_q]),
                 is_record=np.concatenate([np.zeros(nb, np.uint8), np.ones(len(length), np.uint8)]))
    ff = np.concatenate([b["fail_flag"], np.zeros(len(length), np.uint8)])
    wt = c_oracle.tables(stats, np.concatenate([b["channel"], channel]), np.concatenate([b["mux"], mux]),
                         np.concatenate([b["time"], t]), ff, BASE_CFG)
    assert_tables_equal(tables, wt, "mixed")


@pytest.mark.parametrize("mode", [0, 1])
def test_sub_second_ticks(qc, c_oracle, mode):
    """Times in milliseconds: bin_seconds = 3 600 000 ticks, 64-bit tz offset and reference time."""
    tick = 1000
    cfg = dict(BASE_CFG, n_mux=1, time_mode=mode, n_channels=16, bin_seconds=3600 * tick, tz_offset_s=3600 * tick,
               ref_time_s=1_528_000_000 * tick)
    length, mean_q, gc, channel, mux, t = _record_columns(81, 3000, 16, hours=47, tick=tick)
    with qc.QCContext(device=0, **cfg) as ctx:
        ctx.submit_records(length, mean_q, channel, mux, t, gc)
        tables = ctx.finalize()
    assert_tables_equal(tables, c_oracle.tables(_record_stats(length, mean_q, gc), channel, mux, t, None, cfg))


def test_fastq_text_to_tables(qc, c_oracle):
    """FASTQ text -> ppqc_fastq_pack -> CSR -> GPU: per-record GC of fastq_reader (lib/seq_routines.py:40-53) and the
    header-derived channel / start time."""
    from oracle import pypore_oracle as po
    from pypore_amd import fastq_io
    rng = np.random.default_rng(91)
    recs = []
    for i in range(400):
        L = int(rng.integers(0, 3000))
        s = bytes(rng.choice(np.frombuffer(b"ACGTNacgtSs", np.uint8), L))
        q = bytes(rng.integers(33, 80, L, dtype=np.uint8))
        sec = int(rng.integers(0, 47 * 3600))
        hdr = "r%d runid=x read=%d ch=%d start_time=2018-06-03T%02d:%02d:%02dZ" % (
            i, i, 1 + i % 512, (sec // 3600) % 24, (sec // 60) % 60, sec % 60)
        recs.append((hdr.encode(), s, q))
    text = b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in recs)
    p = fastq_io.pack_fastq(text)
    ref = int(p["start_time"].min())
    cfg = dict(BASE_CFG, ref_time_s=ref)
    mux = np.ones(p["n_reads"], np.uint8)
    with qc.QCContext(device=0, **cfg) as ctx:
        ctx.submit(p["bases"], p["quals"], p["offsets"], p["channel"], mux, p["start_time"])
        got = ctx.read_stats()
        tables = ctx.finalize()
    gc_pct = [g * 100.0 / l if l else 0.0 for g, l in zip(got["gc"].tolist(), got["length"].tolist())]
    assert gc_pct == po.fastq_reader(text)
    want = c_oracle.per_read(p["bases"], p["quals"], p["offsets"], nthreads=2)
    _check_reads(got, want)
    assert_tables_equal(tables, c_oracle.tables(want, p["channel"], mux, p["start_time"], None, cfg))


def test_n50_radix_select_equals_sort(qc, c_oracle):
    """The multi-GPU N50 path (three histogram / pick rounds) run on one GPU against the LSD-sort N50 and the oracle."""
    rng = np.random.default_rng(17)
    cases = [[2, 2, 2, 3, 3, 4, 8, 8], [10, 20, 30, 40, 50], [5], [0, 0, 0], [7] * 5000, [1, 2_000_000],
             rng.integers(1, 2_000_000, 70_001).tolist(), rng.integers(1, 4000, 300_000).tolist(),
             np.rint(rng.lognormal(8.5, 1.2, 200_000)).astype(np.int64).clip(1, 2_000_000).tolist()]
    for lens in cases:
        R = len(lens)
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
        nb = int(off[-1])
        with qc.QCContext(device=0, **BASE_CFG) as ctx:
            ctx.submit(np.full(nb, ord("A"), np.uint8), np.full(nb, 50, np.uint8), off, np.ones(R, np.uint32),
                       np.ones(R, np.uint8), np.full(R, 1_528_000_000, np.int64))
            ctx.finalize_extremes()
            ctx.finalize_tables()
            ctx.n50_select()
            a = ctx.fetch_tables()
            ctx.finalize_extremes()
            ctx.finalize_tables()
            ctx.finalize_n50()
            b = ctx.fetch_tables()
        assert a["n50"] == b["n50"] == c_oracle.n50(lens), (lens[:8], a["n50"], b["n50"])
        assert_tables_equal(a, b)


def test_table_views_equal_copies(qc):
    """ppqc_finalize_view / ppqc_fetch_tables_view (pointers into the pinned result buffer) give the tables of
    ppqc_finalize / ppqc_fetch_tables, on two runs in a row of one context (the second smaller)."""
    cfg = dict(n_channels=64, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0, tz_offset_s=0,
               ref_time_s=1_528_000_000, fail_q=7.0)
    with qc.QCContext(device=0, **cfg) as ctx:
        for seed, n in ((5, 3000), (6, 700)):
            b = random_batch(seed, n, 1500, n_channels=64)
            ctx.reset()
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
            want = ctx.finalize()
            ctx.reset()
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
            view = ctx.finalize(copy=False)
            view2 = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in view.items()}
            assert_tables_equal(view2, want)
            again = ctx.fetch_tables(copy=False)
            assert all(np.array_equal(again[k], want[k]) for k in want)
            assert int(want["totals"][0]) == n


def test_boundary_fuzz_both_tile_kernels():
    """tools/fuzz_parity.py, 12 cases: length distributions that stress read boundaries (runs of tiny reads, reads
    ending on vector / span / tile edges, reads of several tiles) through both tile-kernel instantiations and three
    tile sizes, per-read integers bit-exact against the C oracle."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "12", "4242"],
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_malformed_device_csr_fails_loudly_without_faulting(qc):
    """A device batch cannot be validated on the host; bad offsets must neither fault nor pass."""
    import torch
    from pypore_amd._lib import PPQCError
    nb = 200_000
    tb = torch.full((nb + 64,), ord("G"), dtype=torch.uint8, device="cuda")
    tq = torch.full((nb + 64,), 60, dtype=torch.uint8, device="cuda")
    meta = [torch.ones(4, dtype=torch.int32, device="cuda"), torch.ones(4, dtype=torch.uint8, device="cuda"),
            torch.full((4,), 1_528_000_000, dtype=torch.int64, device="cuda")]
    for offs in ([0, 100_000, 50_000, 150_000, nb],          # not monotone
                 [0, 10, 20, 30, nb + (1 << 40)],              # runs far past the data
                 [16, 100, 200, 300, nb]):                     # does not start at 0
        with qc.QCContext(device=0, **BASE_CFG) as ctx:
            d_off = torch.tensor(offs, dtype=torch.int64, device="cuda")
            ctx.submit_device(4, nb, tb.data_ptr(), tq.data_ptr(), d_off.data_ptr(), *[m.data_ptr() for m in meta])
            with pytest.raises(PPQCError, match="malformed CSR"):
                ctx.read_stats()
            with pytest.raises(PPQCError, match="malformed CSR"):
                ctx.finalize()
            ctx.reset()                                        # the context stays usable
            good = torch.tensor([0, 10, 20, 30, nb], dtype=torch.int64, device="cuda")
            ctx.submit_device(4, nb, tb.data_ptr(), tq.data_ptr(), good.data_ptr(), *[m.data_ptr() for m in meta])
            t = ctx.finalize()
            assert int(t["totals"][1]) == nb and int(t["totals"][3]) == nb
