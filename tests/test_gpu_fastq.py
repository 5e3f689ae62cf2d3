"""FASTQ text tokenised on the GPU (ppqc_submit_fastq / ppqc_submit_fastq_device) against the host packer + CSR path
and the oracle: same per-read records and tables, bit for bit, for LF / CRLF text, with and without a final newline,
repeated captions, reads longer than a tile, empty reads; malformed text fails loudly."""
import numpy as np
import pytest

from util import assert_tables_equal

pytestmark = pytest.mark.gpu

CFG = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0, tz_offset_s=0,
           ref_time_s=1_527_984_000, fail_q=7.0)
ALPH = np.frombuffer(b"ACGTACGTNnacgtSs", np.uint8)


WAYS = {"second_pass": {"fastq_sums": 0, "fastq_one_pass": 0}, "span_sums": {"fastq_sums": 2, "fastq_one_pass": 0},
        "auto": {"fastq_sums": 1}, "one_pass": {"fastq_one_pass": 1}, "one_pass_3_regions": {"fastq_one_pass": 1, "fq1_regions": 3}}


@pytest.fixture(autouse=True, params=list(WAYS), ids=list(WAYS))
def line_sums_from(request, monkeypatch):
    """Every test runs every way the library has of turning text into per-read sums: the line-mode tile kernel's second
    pass over the text (ppqc_set_tuning fastq_sums = 0), the span sums of the scan pass put together per line (2), the
    one-pass kernels of ppqc_fastq1.hip (fastq_one_pass = 1; with 3 regions a wave walks many chunks and the records
    straddle regions), and the library's own choice (by the mean line length of the context's previous text)."""
    from pypore_amd import qc
    monkeypatch.setattr(qc.QCContext, "DEFAULT_TUNING", dict(WAYS[request.param]))
    return request.param


def _text(seed, n, crlf=False, trailing=True, plus_title=False, lens=None, fields=True):
    rng = np.random.default_rng(seed)
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(lens[i]) if lens is not None else int(rng.choice([0, 1, 15, 16, 17, 1023, 1024, 1025, int(rng.integers(1, 9000))]))
        seq = bytes(rng.choice(ALPH, L))
        qual = bytes(rng.integers(33, 127, L, dtype=np.uint8))
        sec = int(rng.integers(0, 47 * 3600))
        title = b"read%d runid=ab read=%d" % (i, i)
        if fields:
            title += b" ch=%d start_time=2018-06-%02dT%02d:%02d:%02dZ" % (
                int(rng.integers(1, 513)), 3 + sec // 86400, (sec // 3600) % 24, (sec // 60) % 60, sec % 60)
        if i % 5 == 0:
            title += b"  "                      # trailing blanks are stripped like every line end
        out.append(b"@" + title + nl + seq + nl + b"+" + (title.rstrip() if plus_title else b"") + nl + qual + nl)
    t = b"".join(out)
    return t if trailing else t[:-len(nl)]


def _took(ctx, way):
    """With the one-pass kernels forced, a text they are made for must not have been handed to the general path."""
    if way.startswith("one_pass"):
        assert ctx.path_count("fastq_one_pass") == 1 and ctx.path_count("fastq_declined") == 0


def _host_path(qc, c_oracle, text):
    from pypore_amd import fastq_io
    p = fastq_io.pack_fastq(text)
    mux = np.ones(p["n_reads"], np.uint8)
    want = c_oracle.per_read(p["bases"], p["quals"], p["offsets"], nthreads=2)
    tables = c_oracle.tables(want, p["channel"], mux, p["start_time"], None, CFG)
    return p, want, tables


def _check(got, want):
    for k in ("length", "gc", "n", "qsum"):
        assert np.array_equal(got[k], want[k]), k
    assert [(int(h) << 32) + int(l) for h, l in zip(got["err_hi"], got["err_lo"])] == [int(e) for e in want["err"]]
    np.testing.assert_allclose(got["mean_q"], want["mean_q"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("kw", [dict(), dict(crlf=True), dict(trailing=False), dict(crlf=True, trailing=False),
                                dict(plus_title=True)])
def test_gpu_tokenizer_equals_host_packer(qc_mod, c_oracle, kw):
    text = _text(11, 700, **kw)
    p, want, wt = _host_path(qc_mod, c_oracle, text)
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == p["n_reads"] == 700
        got = ctx.read_stats()
        tables = ctx.finalize()
    _check(got, want)
    assert_tables_equal(tables, wt, str(kw))


def test_long_reads_many_files_both_slots_and_device_variant(qc_mod, c_oracle):
    import torch
    lens = [200_003, 65536, 65537, 0, 131072, 5, 70_000] + [300] * 50
    texts = [_text(21 + k, len(lens), lens=lens, crlf=bool(k % 2)) for k in range(5)]
    whole = []
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        for k, t in enumerate(texts):
            assert ctx.submit_fastq(t, slot=k % 2) == len(lens)
        got = ctx.read_stats()
        tables = ctx.finalize()
        # the same texts, already resident in HBM
        ctx.reset()
        keep = []
        for t in texts:
            d = torch.from_numpy(np.frombuffer(t, np.uint8).copy()).cuda()
            pad = torch.zeros(d.numel() + 64, dtype=torch.uint8, device="cuda")
            pad[: d.numel()] = d
            keep.append(pad)
            assert ctx.submit_fastq_device(pad.data_ptr(), d.numel()) == len(lens)
        got2 = ctx.read_stats()
        tables2 = ctx.finalize()
    stats, ch, st = [], [], []
    for t in texts:
        p, want, _ = _host_path(qc_mod, c_oracle, t)
        stats.append(want); ch.append(p["channel"]); st.append(p["start_time"])
    want = {k: np.concatenate([s[k] for s in stats]) for k in ("length", "gc", "n", "qsum", "mean_q")}
    want["err"] = [e for s in stats for e in s["err"]]
    _check(got, want)
    _check(got2, want)
    ch, st = np.concatenate(ch), np.concatenate(st)
    wt = c_oracle.tables(want, ch, np.ones(len(ch), np.uint8), st, None, CFG)
    assert_tables_equal(tables, wt)
    assert_tables_equal(tables2, wt)


def test_headers_without_fields_and_empty_text(qc_mod):
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(b"") == 0
        assert ctx.submit_fastq(_text(3, 40, fields=False)) == 40
        t = ctx.finalize()
        assert int(t["totals"][0]) == 40 and int(t["totals"][7]) == 40       # channel 0 = bad channel


@pytest.mark.parametrize("bad,where", [
    (b"@r\nACGT\n+\nII\n", "finalize"),                 # lengths differ
    (b"r\nACGT\n+\nIIII\n", "finalize"),                # no '@'
    (b"@r\nACGT\n-\nIIII\n", "finalize"),               # no '+'
    (b"@r\nACGT\n+other\nIIII\n", "finalize"),          # captions differ
    (b"@r\nACGT\n+\n", "submit"),                       # truncated: 3 lines
    (b"@r\nACGT\n+\nIIII\n\n@s\nAC\n+\nII\n", "submit"),  # blank line between records: 9 lines
])
def test_malformed_text_fails_loudly(qc_mod, bad, where):
    from pypore_amd._lib import PPQCError
    good = _text(5, 10)
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        if where == "submit":
            with pytest.raises(PPQCError, match="multiple of 4"):
                ctx.submit_fastq(good + bad)
        else:
            ctx.submit_fastq(good + bad)
            with pytest.raises(PPQCError, match="record 10"):
                ctx.finalize()
            with pytest.raises(PPQCError, match="malformed FASTQ"):
                ctx.read_stats()
        ctx.reset()
        assert ctx.submit_fastq(good) == 10 and int(ctx.finalize()["totals"][0]) == 10


def test_invalid_quality_bytes_are_counted_like_the_host_packer(qc_mod):
    from pypore_amd import fastq_io
    rng = np.random.default_rng(8)
    recs = []
    for i in range(300):
        L = int(rng.integers(1, 4000))
        q = rng.integers(33, 127, L, dtype=np.uint8)
        k = rng.random(L) < 0.01
        q[k] = rng.choice(np.array([0, 1, 8, 14, 31, 127, 128, 200, 255], np.uint8), int(k.sum()))
        # bases full of bytes that are invalid as qualities must not count; neither must CR LF line ends
        s = rng.choice(np.array([65, 67, 71, 84, 1, 200, 127], np.uint8), L)
        recs.append(b"@r%d ch=3\r\n" % i + bytes(s) + b"\r\n+\r\n" + bytes(q) + b"\r\n")
    text = b"".join(recs)
    want = fastq_io.pack_fastq(text)["n_bad_qual"]
    assert want > 100
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        ctx.submit_fastq(text)
        e = ctx.input_errors()
        assert e == dict(bad_csr_rows=0, bad_fastq_records=0, first_bad_record=None, bad_quality_bytes=want,
                         misplaced_whitespace=0)
        ctx.finalize()                 # reported, not fatal (the bytes are clamped as in ppqc_submit)
        ctx.reset()
        assert ctx.input_errors()["bad_quality_bytes"] == 0


def test_dense_text_takes_the_second_text_pass(qc_mod, c_oracle):
    """The first pass over the text notes at most 32 vectors per 8 KiB chunk that hold a line end or an invalid quality
    byte, and the line starts are placed from those notes.  Reads of a few bases give a line end in every other vector:
    the lists overflow and the line starts come from a second pass over the text instead -- same records, tables and
    invalid-quality-byte count as the host packer either way.  A sparser text on the same context then goes back to the
    notes."""
    from pypore_amd import fastq_io
    rng = np.random.default_rng(12)
    recs = []
    for i in range(30_000):
        L = int(rng.integers(0, 6))
        q = rng.integers(33, 127, L, dtype=np.uint8)
        if L and i % 50 == 0:
            q[0] = 200                                      # invalid as a quality
        recs.append(b"@r%d ch=%d\n" % (i, 1 + i % 512) + bytes(rng.choice(ALPH, L)) + b"\n+\n" + bytes(q) + b"\n")
    text = b"".join(recs)
    assert len(text) / (4 * len(recs)) < 20
    p, want, _ = _host_path(qc_mod, c_oracle, text)
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == len(recs)
        _check(ctx.read_stats(), want)
        assert ctx.input_errors()["bad_quality_bytes"] == p["n_bad_qual"] > 100
        ctx.reset()
        # the same text again: by its own choice the library now knows the lines are short and only counts in the scan
        assert ctx.submit_fastq(text) == len(recs)
        _check(ctx.read_stats(), want)
        assert ctx.input_errors()["bad_quality_bytes"] == p["n_bad_qual"]
        ctx.reset()
        sparse = _text(13, 500)
        p2, want2, wt2 = _host_path(qc_mod, c_oracle, sparse)
        assert ctx.submit_fastq(sparse) == 500
        _check(ctx.read_stats(), want2)
        assert_tables_equal(ctx.finalize(), wt2)


def test_lines_that_end_on_span_group_and_tile_edges_and_odd_headers(qc_mod, c_oracle, line_sums_from):
    """The line-mode tile kernel takes four 1 KiB spans at once when the line covers them: lines that end exactly on a
    4 KiB group edge, a 1 KiB span edge or a 64 KiB tile edge, and lines a few bytes off them.  The header kernel works on
    240-byte windows in LDS: longer headers (the one-lane walk), keys at the very start, behind a tab, inside other
    words, repeated, cut short, and fields that are no numbers."""
    rng = np.random.default_rng(5)
    heads = [b"ch=7 start_time=2018-06-03T01:02:03Z",                       # key right behind the '@'
             b"r xch=9 ch=12\tstart_time=2018-06-04T10:00:00Z ch=99",       # first standalone ch= wins; tab before the key
             b"r start_time=2018-06-03T01:02 ch=3",                         # time cut short: no start time
             b"r ch=abc start_time=2018-13-03T01:02:03Z",                   # no digits: channel 0; month 13: no time
             b"r " + b"x" * 300 + b" ch=44 start_time=2018-06-03T05:00:00Z",  # longer than the LDS window
             b"r " + b"y" * 219 + b" ch=45 start_time=2018-06-03T06:00:00Z",  # just inside it
             b"r " + b"y" * 225 + b" ch=46 start_time=2018-06-03T06:00:00Z   ",
             b"r ch=47 start_time=2018-06-03 07:00:00"]                     # blank instead of 'T'
    out, pos = [], 0
    edges = [4096, 1024, 65536, 4096, 8192, 4096 * 3, 65536 * 2]
    for i in range(160):
        title = heads[i % len(heads)] if i % 3 == 0 else b"read%d ch=%d start_time=2018-06-03T%02d:00:00Z" % (i, 1 + i % 512, i % 24)
        h = len(title) + 2                                  # '@' + title + '\n'
        E = edges[i % len(edges)]
        d = int(rng.integers(-2, 3)) if i % 4 == 3 else 0   # some lines miss the edge by a byte or two
        if i % 2 == 0:                                      # the sequence line ends on the edge
            L = (-(pos + h + 1)) % E + d
        else:                                               # the quality line does
            if (pos + h + 4) % 2:
                title += b" "; h += 1
            L = ((-(pos + h + 4)) % E) // 2 + d
        L = max(L, 0) + (E if L < 40 else 0)
        seq = bytes(rng.choice(ALPH, L)); qual = bytes(rng.integers(33, 127, L, dtype=np.uint8))
        rec = b"@" + title + b"\n" + seq + b"\n+\n" + qual + b"\n"
        out.append(rec); pos += len(rec)
    text = b"".join(out)
    p, want, wt = _host_path(qc_mod, c_oracle, text)
    assert p["n_reads"] == 160
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == 160
        got = ctx.read_stats()
        tables = ctx.finalize()
        _took(ctx, line_sums_from)
    _check(got, want)
    assert_tables_equal(tables, wt)      # channels and start times: the host packer's, through the tables
    assert int(tables["totals"][7]) == int(np.sum((p["channel"] < 1) | (p["channel"] > 512))) > 0


def test_header_fuzz_against_the_host_packer(qc_mod, c_oracle):
    """Random token soups in the header (keys whole, cut, glued to other words, repeated, behind tabs, at every offset,
    fields running into the line end, LF and CRLF) must give the channels and start times of the host packer -- through the
    per-channel and per-hour tables, with one base per read so that every table cell counts reads."""
    rng = np.random.default_rng(77)
    toks = [b"ch=%d", b"xch=%d", b"ch=", b"ch", b"c", b"h=%d", b"ch=%dq", b"start_time=2018-06-03T%02d:10:00Z",
            b"start_time=2018-06-04 %02d:00:00", b"start_time=2018-06-03T%02d", b"start_time=", b"start_tim=2018-06-03T%02d:00:00Z",
            b"xstart_time=2018-06-03T%02d:00:00Z", b"runid=%032x", b"read=%d", b"s", b"st", b"", b"start_time=2018-06-03T%02d:61:00Z"]
    recs = []
    for i in range(3000):
        parts = []
        for _ in range(int(rng.integers(0, 9))):
            t = toks[int(rng.integers(0, len(toks)))]
            if b"%" in t:
                t = t % (int(rng.integers(0, 24)) if b"start" in t else int(rng.integers(0, 600)))
            parts.append(t)
            parts.append(rng.choice([b" ", b"\t", b"  ", b" \t"]))
        title = b"".join(parts)
        if i % 3 == 0:
            title = title.rstrip()          # a field that runs into the line end
        nl = b"\r\n" if i % 7 == 0 else b"\n"
        recs.append(b"@" + title + nl + b"G" + nl + b"+" + nl + b"I" + nl)
    text = b"".join(recs)
    p, want, wt = _host_path(qc_mod, c_oracle, text)
    assert p["n_reads"] == 3000
    assert len(set(p["channel"].tolist())) > 200 and len(set(p["start_time"].tolist())) > 20
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == 3000
        got = ctx.read_stats()
        tables = ctx.finalize()
    _check(got, want)
    assert_tables_equal(tables, wt)


def test_a_run_of_short_reads_inside_long_ones(qc_mod, c_oracle):
    """A few chunks of the text hold more line ends than their notes (32 vectors per 8 KiB): only those chunks are read
    again for their line starts, the rest of the text is placed from the notes -- and with many such chunks the whole
    text takes the second pass.  Same records and tables as the host packer in every mix."""
    for seed, lens in ((31, [9000] * 30 + [3] * 400 + [12000] * 30 + [0] * 200 + [70000, 5]),
                       (32, [2] * 900 + [20000] * 3),
                       (33, [15000] * 20 + [1] * 40 + [15000] * 20)):
        text = _text(seed, len(lens), lens=lens, crlf=bool(seed % 2))
        p, want, wt = _host_path(qc_mod, c_oracle, text)
        with qc_mod.QCContext(device=0, **CFG) as ctx:
            assert ctx.submit_fastq(text) == len(lens)
            got = ctx.read_stats()
            tables = ctx.finalize()
        _check(got, want)
        assert_tables_equal(tables, wt, str(seed))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PPQC_FUZZ_SEEDS", "10"))))   # more for a longer run
def test_fuzzed_texts_around_span_edges(qc_mod, c_oracle, seed, line_sums_from):
    """Random texts aimed at the span sums: lines that start or end on, just before or just behind a 1 KiB span edge,
    lines inside one span, empty lines, lines of many spans; quality lines of nothing but bytes at or above '@' (the scan
    pass then takes them for sequence and the line sums must come from the text), sequence lines with bytes below '@',
    invalid bytes inside lines (a span without sums inside a line), headers longer than a span, LF / CRLF, with and without the
    last line end.  Both ways of making the line sums against the host packer and the oracle."""
    rng = np.random.default_rng(1000 + seed)
    crlf = bool(seed % 2)
    nl = b"\r\n" if crlf else b"\n"
    out, pos = [], 0
    for i in range(int(rng.integers(150, 400))):
        kind = int(rng.integers(0, 8))
        if kind == 0:
            L = int(rng.integers(0, 3))
        elif kind == 1:
            L = int(rng.integers(1, 900))
        elif kind == 2:
            L = 1024 * int(rng.integers(1, 6)) + int(rng.integers(-3, 4))
        elif kind == 3:
            L = int(rng.integers(3000, 40000))
        elif kind == 4:
            L = int(rng.integers(60000, 140000))
        else:
            L = int(rng.integers(900, 6000))
        title = b"r%d ch=%d start_time=2018-06-03T%02d:00:00Z" % (i, int(rng.integers(1, 513)), int(rng.integers(0, 24)))
        if kind == 5:
            title += b" " + b"k" * int(rng.integers(1000, 2500))
        if kind in (2, 6):      # put the start of the sequence line on (or next to) a span edge
            pad = (-(pos + 1 + len(title) + len(nl))) % 1024 + int(rng.integers(-2, 3))
            if pad > 2:
                title += b" " + b"p" * (pad - 1)
        seq = rng.choice(ALPH, L)
        hiq = int(rng.integers(0, 3)) == 0
        qual = rng.integers(64 if hiq else 33, 127, L, dtype=np.uint8)
        if L > 50 and kind == 7:
            seq = seq.copy()
            seq[rng.integers(0, L, 3)] = np.frombuffer(b"*-.", np.uint8)       # bytes below '@' inside a sequence
            qual[rng.integers(0, L - 1, 2)] = 31                                # invalid bytes inside a quality line
        rec = b"@" + title + nl + bytes(seq) + nl + b"+" + nl + bytes(qual) + nl
        out.append(rec); pos += len(rec)
    text = b"".join(out)
    if seed % 3 == 0:
        text = text[:-len(nl)]
    p, want, wt = _host_path(qc_mod, c_oracle, text)
    assert p["n_reads"] == len(out)
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == len(out)
        got = ctx.read_stats()
        tables = ctx.finalize()
        _took(ctx, line_sums_from)
    _check(got, want)
    assert_tables_equal(tables, wt, str(seed))


def _host_verdict(text):
    """True iff the host packer (ppqc_fastq_scan / ppqc_fastq_pack, which follow SeqIO) takes the text without complaint:
    a malformed record raises, a quality byte outside '!'..'~' is counted and makes the CLI raise."""
    from pypore_amd import fastq_io
    try:
        return fastq_io.pack_fastq(text)["n_bad_qual"] == 0
    except ValueError:
        return False


def _gpu_verdict(qc_mod, text):
    from pypore_amd._lib import PPQCError
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        try:
            ctx.submit_fastq(text)
            e = ctx.input_errors()
            ctx.read_stats()
            ctx.finalize()
        except PPQCError as err:
            assert "FASTQ" in str(err)
            return False, str(err)
        assert e["bad_fastq_records"] == 0 and e["misplaced_whitespace"] == 0
        return e["bad_quality_bytes"] == 0, ""


@pytest.mark.parametrize("rec", [
    b"@r ch=1\nAC GT\n+\nIIIII\n",        # a blank inside the sequence (SeqIO: "Whitespace is not allowed in the sequence")
    b"@r ch=1\nAC\tGT\n+\nIIIII\n",       # a tab
    b"@r ch=1\n ACGT\n+\nIIIII\n",        # at its start
    b"@r ch=1\nAC\rGT\n+\nIIIII\n",       # a bare CR is a line end to SeqIO (universal newlines): not supported
    b"@r ch=1\nACGT\n+\nII\rI\n",         # ... inside the quality line
    b"@r\rch=1\nACGT\n+\nIIII\n",         # ... inside the title
    b"@r ch=1\nACGT\n+\nII I\n",          # a blank as a quality
    b"@r ch=1\nACGT\n+\nII\x0bI\n",       # a vertical tab as a quality
    b"@r ch=1\r\nAC GT\r\n+\r\nIIIII\r\n",
])
@pytest.mark.parametrize("n_before", [0, 40])
def test_white_space_where_seqio_refuses_it(qc_mod, rec, n_before):
    """the two cases the header used to list as not checked are failures now, with the host packer's
    verdict; the records around the bad one are fine and the context works again after a reset."""
    from pypore_amd._lib import PPQCError
    good = _text(5, n_before, crlf=b"\r\n" in rec) if n_before else b""
    text = good + rec + good
    assert _host_verdict(good + good) and not _host_verdict(text)
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == 2 * n_before + 1
        assert ctx.input_errors()["misplaced_whitespace"] == 1
        with pytest.raises(PPQCError, match="white-space"):
            ctx.finalize()
        with pytest.raises(PPQCError, match="white-space"):
            ctx.read_stats()
        ctx.reset()
        assert ctx.submit_fastq(good + b"@r \t\x0b x\r\nACGT \r\n+\t\r\nIIII\x0c\r\n") == n_before + 1
        assert ctx.input_errors()["misplaced_whitespace"] == 0      # all of it in the title or stripped at a line end
        assert int(ctx.finalize()["totals"][0]) == n_before + 1


@pytest.mark.parametrize("seed", range(6))
def test_white_space_fuzz_has_the_host_packers_verdict(qc_mod, seed):
    """Texts of short and long records with white-space bytes dropped in at random places -- in front of, inside and at
    the end of each of the four lines -- are accepted or refused exactly as the host packer does (blank, tab, CR
    anywhere; \\v and \\f anywhere but inside a sequence line, where SeqIO takes them for bases and this tokenizer,
    which cannot tell which line a misplaced byte lies in, refuses them: include/pypore_qc.h)."""
    rng = np.random.default_rng(500 + seed)
    agree = {True: 0, False: 0}
    for case in range(40):
        crlf = bool(rng.integers(0, 2))
        nl = b"\r\n" if crlf else b"\n"
        recs = []
        for i in range(int(rng.integers(1, 30))):
            L = int(rng.choice([1, 2, 17, 300, 1100, 5000, 20000]))
            title = b"r%d ch=%d start_time=2018-06-03T%02d:00:00Z" % (i, int(rng.integers(1, 513)), int(rng.integers(0, 24)))
            recs.append([b"@" + title, bytes(rng.choice(ALPH, L)), b"+" + (title if rng.integers(0, 4) == 0 else b""),
                         bytes(rng.integers(33, 127, L, dtype=np.uint8))])
        for _ in range(int(rng.integers(0, 3))):          # 0, 1 or 2 bytes dropped in
            r = recs[int(rng.integers(0, len(recs)))]
            k = int(rng.integers(0, 4))
            ch = bytes([int(rng.choice([32, 9, 13] if k == 1 else [32, 9, 13, 11, 12]))])
            where = int(rng.integers(0, 3))
            at = 0 if where == 0 else (len(r[k]) if where == 1 else int(rng.integers(0, len(r[k]) + 1)))
            r[k] = r[k][:at] + ch + r[k][at:]
        text = nl.join(nl.join(r) for r in recs) + (nl if rng.integers(0, 2) else b"")
        want = _host_verdict(text)
        got, msg = _gpu_verdict(qc_mod, text)
        assert got == want, (seed, case, msg, text[:200])
        agree[want] += 1
    assert agree[True] >= 5 and agree[False] >= 5


_BIG = {}


def _big_text():
    """0.4 GB of text, made once for the three ways the test below runs."""
    if not _BIG:
        rng = np.random.default_rng(99)
        lens = np.clip(np.rint(rng.lognormal(8.5172, 1.1774, 20_000)), 1, 2_000_000).astype(np.int64)
        nb = int(lens.sum())
        hb = rng.choice(np.frombuffer(b"ACGTACGTACGTN", np.uint8), nb).tobytes()
        hq = rng.integers(34, 70, nb, dtype=np.uint8).tobytes()
        off = np.concatenate([[0], np.cumsum(lens)])
        _BIG["text"] = b"".join(b"@r%d runid=f read=%d ch=%d start_time=2018-06-03T%02d:%02d:00Z\n%s\n+\n%s\n" % (
            i, i, 1 + i % 512, i % 24, i % 60, hb[off[i]:off[i + 1]], hq[off[i]:off[i + 1]]) for i in range(len(lens)))
        _BIG["n"], _BIG["nb"] = len(lens), nb
    return _BIG["text"], _BIG["n"], _BIG["nb"]


def test_a_text_of_long_reads_at_size(qc_mod, c_oracle, line_sums_from):
    """0.4 GB of text, 20 000 reads with the length distribution of BASELINE config 2 (lognormal, mean 10 kb, up to
    hundreds of kb): every record and table against the host packer + C oracle, however the line sums are made."""
    text, n_reads, nb = _big_text()
    if "host" not in _BIG:
        _BIG["host"] = _host_path(qc_mod, c_oracle, text)
    p, want, wt = _BIG["host"]
    assert p["n_reads"] == n_reads and int(want["length"].sum()) == nb
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == n_reads
        got = ctx.read_stats()
        tables = ctx.finalize()
        _took(ctx, line_sums_from)
    _check(got, want)
    assert_tables_equal(tables, wt)
    assert int(tables["totals"][1]) == nb


def test_a_text_of_short_reads_at_size(qc_mod, c_oracle, line_sums_from):
    """0.25 GB of 500-base reads (the regime the one-pass kernels are for: BASELINE's FASTQ leg at a mean of 500), LF
    and CRLF halves, every record and table against the host packer + C oracle; by its own choice the library takes the
    one-pass kernels for the second text of a context (the first one tells it how long the lines are)."""
    if "short" not in _BIG:
        rng = np.random.default_rng(123)
        lens = np.clip(np.rint(rng.lognormal(np.log(500) - 0.32, 0.8, 220_000)), 1, 60_000).astype(np.int64)
        nb = int(lens.sum())
        hb = rng.choice(np.frombuffer(b"ACGTACGTACGTN", np.uint8), nb).tobytes()
        hq = rng.integers(33, 90, nb, dtype=np.uint8).tobytes()
        off = np.concatenate([[0], np.cumsum(lens)])
        half = len(lens) // 2
        text = b"".join(b"@r%d runid=f read=%d ch=%d start_time=2018-06-03T%02d:%02d:00Z%s%s%s+%s%s%s" % (
            i, i, 1 + i % 512, i % 24, i % 60, nl, hb[off[i]:off[i + 1]], nl, nl, hq[off[i]:off[i + 1]], nl)
            for i in range(len(lens)) for nl in ((b"\n" if i < half else b"\r\n"),))
        _BIG["short"] = (text, len(lens), nb, _host_path(qc_mod, c_oracle, text))
    text, n_reads, nb, (p, want, wt) = _BIG["short"]
    assert p["n_reads"] == n_reads and int(want["length"].sum()) == nb
    with qc_mod.QCContext(device=0, **CFG) as ctx:
        assert ctx.submit_fastq(text) == n_reads
        got = ctx.read_stats()
        tables = ctx.finalize()
        _took(ctx, line_sums_from)
        _check(got, want)
        assert_tables_equal(tables, wt)
        ctx.reset()
        assert ctx.submit_fastq(text) == n_reads         # again: now the context knows the lines are short
        if line_sums_from == "auto":
            assert ctx.path_count("fastq_one_pass") == 1
        assert_tables_equal(ctx.finalize(), wt)


TAILS = [dict(), dict(fastq_extremes=0), dict(n50_select=0), dict(fastq_extremes=0, n50_select=0), dict(n50_select=1)]


@pytest.mark.parametrize("tail", TAILS, ids=lambda t: ",".join("%s=%d" % kv for kv in t.items()) or "default")
def test_the_tail_of_a_step_straight_behind_the_texts(qc_mod, c_oracle, tail):
    """ppqc_finalize with nothing between it and the texts (no ppqc_sync: the extremes kernel of the last text may still
    run on the auxiliary stream, the tables take the largest length from the sort gate, the N50 comes from the radix
    select) against the same with the extremes in line and / or the N50 by sort, in runs of one text, of several texts,
    of a CSR batch and texts in either order (the gate then holds ORs, the tables wait for the extremes), of texts whose
    reads have no valid channel (every sort key 0), and a second finalize of the same reads."""
    from pypore_amd import fastq_io
    rng = np.random.default_rng(77)
    lens_a = [int(x) for x in rng.integers(1, 700, 900)] + [40_000, 0, 3]
    lens_b = [int(x) for x in rng.integers(100, 300, 1500)]
    lens_c = [70_000, 5, 131_073] + [int(x) for x in rng.integers(1, 5000, 300)]
    texts = [_text(201, len(lens_a), lens=lens_a), _text(202, len(lens_b), lens=lens_b, crlf=True),
             _text(203, len(lens_c), lens=lens_c)]
    packs = [fastq_io.pack_fastq(t) for t in texts]

    def expect(order):
        stats = [c_oracle.per_read(packs[k]["bases"], packs[k]["quals"], packs[k]["offsets"], nthreads=2) for k in order]
        want = {k: np.concatenate([s[k] for s in stats]) for k in ("length", "gc", "n", "qsum", "mean_q")}
        want["err"] = [e for s in stats for e in s["err"]]
        ch = np.concatenate([packs[k]["channel"] for k in order])
        st = np.concatenate([packs[k]["start_time"] for k in order])
        return c_oracle.tables(want, ch, np.ones(len(ch), np.uint8), st, None, CFG)

    def as_batch(ctx, k):
        p = packs[k]
        ctx.submit(p["bases"], p["quals"], p["offsets"], p["channel"], np.ones(p["n_reads"], np.uint8), p["start_time"])

    with qc_mod.QCContext(device=0, **CFG) as ctx:
        for name, value in tail.items():
            ctx.set_tuning(name, value)
        for plan in ([("t", 0)], [("t", 1), ("t", 0), ("t", 2)], [("b", 2), ("t", 1)], [("t", 1), ("b", 0)],
                     [("t", 0), ("b", 1), ("t", 2)]):
            ctx.reset()
            for kind, k in plan:
                if kind == "t":
                    assert ctx.submit_fastq(texts[k], slot=k % 2) == packs[k]["n_reads"]
                else:
                    as_batch(ctx, k)
            wt = expect([k for _, k in plan])
            assert_tables_equal(ctx.finalize(), wt, str(plan))
            assert_tables_equal(ctx.finalize(), wt, "again " + str(plan))   # the same reads once more, without a reset
        # no read with a valid channel: every key is 0, the gate stays 0, N50 0
        bare = _text(204, 300, lens=[int(x) for x in rng.integers(1, 400, 300)], fields=False)
        pb = fastq_io.pack_fastq(bare)
        sb = c_oracle.per_read(pb["bases"], pb["quals"], pb["offsets"], nthreads=2)
        wb = c_oracle.tables(sb, pb["channel"], np.ones(300, np.uint8), pb["start_time"], None, CFG)
        ctx.reset()
        assert ctx.submit_fastq(bare) == 300
        got = ctx.finalize()
        assert_tables_equal(got, wb, "no channels")
        assert int(got["n50"]) == 0


def test_the_finalize_kernel_launched_before_the_line_count_is_known(qc_mod, c_oracle, line_sums_from):
    """From its second text on a context launches the one-pass finalize kernel straight behind the scan, sized by the last
    text's line length; the kernel reads the record count on the device and leaves a text alone that the host will not
    commit as it stands.  Texts like the last one (the kernel runs), a text with far more records than the guess and the
    arrays' room (the kernel declines, the host grows the arrays and launches again), a text the scan declines (short
    lines), a text whose lines are no records (an error, nothing committed) -- each against the host packer."""
    if not line_sums_from.startswith("one_pass"):
        pytest.skip("the one-pass kernels")
    long_lens = [5000] * 20
    mid_lens = [250 + (i % 7) for i in range(3000)]
    tiny_lens = [30] * 600
    t_long, t_long2 = _text(301, len(long_lens), lens=long_lens), _text(302, len(long_lens), lens=long_lens, crlf=True)
    t_mid, t_tiny = _text(303, len(mid_lens), lens=mid_lens), _text(304, len(tiny_lens), lens=tiny_lens)
    for early in (1, 0):
        with qc_mod.QCContext(device=0, **CFG) as ctx:
            ctx.set_tuning("fq1_early", early)
            plan = [t_long, t_long2, t_mid, t_tiny, t_long]
            for t in plan:
                ctx.submit_fastq(t)
            with pytest.raises(Exception, match="multiple of 4"):
                ctx.submit_fastq(t_long + b"@stray\n")
            ctx.submit_fastq(t_long2)
            plan.append(t_long2)
            n_early = ctx.path_count("fastq_early")
            # the second text and the two after the tiny one ran early; the third found too little room, the tiny one
            # was declined by the scan
            assert n_early == (3 if early else 0), n_early
            assert ctx.path_count("fastq_declined") == 1
            got = ctx.read_stats()
            tables = ctx.finalize()
        stats, ch, st = [], [], []
        for t in plan:
            p, want, _ = _host_path(qc_mod, c_oracle, t)
            stats.append(want); ch.append(p["channel"]); st.append(p["start_time"])
        want = {k: np.concatenate([s[k] for s in stats]) for k in ("length", "gc", "n", "qsum", "mean_q")}
        want["err"] = [e for s in stats for e in s["err"]]
        _check(got, want)
        ch, st = np.concatenate(ch), np.concatenate(st)
        assert_tables_equal(tables, c_oracle.tables(want, ch, np.ones(len(ch), np.uint8), st, None, CFG), "early=%d" % early)
