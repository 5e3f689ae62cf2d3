"""Host drop-in (pypore_amd/seq_routines.py, seqstats.py, loaders, FASTQ writer) on the CPU: its table-fed
FastStats / result_plotting data parts against the oracle's reference-shaped layer, the CLI surface against
lib/seqstats.py, the loaders on stand-in HDF5 files.  (Tables come from the oracle here; the GPU supplies them in
tests/test_gpu_cli.py.)"""
import gzip
import json
import os
import time

import numpy as np
import pytest

from oracle import pypore_oracle as po
from pypore_amd import seq_routines as sq, seqstats, fast5_reader as f5r, multi_read_fast5_reader as mf5r, fastq_out
from util import random_batch
from golden.make_golden import toy_batch, toy_reads, TOY_CFG, REF
import fake_h5

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def _ref_shaped(b, st, n_ch, tz, ref):
    tuples = []
    for i in range(len(b["channel"])):
        a, e = int(b["offsets"][i]), int(b["offsets"][i + 1])
        tuples.append(po.get_content_tuple(int(b["channel"][i]), int(b["mux"][i]), int(b["time"][i]),
                                           int(st["length"][i]), float(st["mean_q"][i]),
                                           po.GC(b["bases"][a:e].tobytes()), fail=bool(b["fail_flag"][i])))
    return po.build_channel_dict(tuples, float(ref), n_channels=n_ch, tz_offset_s=tz)


def test_toy_run_summary_table_row_hand_derived():
    b = toy_batch()
    st = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
    t = po.build_tables(st, b["channel"], b["mux"], b["time"], b["fail_flag"], TOY_CFG)
    cd = sq.channel_dict_from_tables(t)
    d = sq.faststats_data(cd)
    want = KAT["toy_run"]
    assert d["table"] == want["table"] == [10, 8, 2, 7100.0, 82.3, 100, 3000]
    assert d["pie"] == want["pie"] and d["FailSum"] == want["FailSum"] and d["PassSum"] == want["PassSum"]
    for h, (r, bs, m) in want["hours"].items():
        assert d["HoursDict"][int(h)] == [r, bs, m]
    for c, k in (("1", "fail_percent_ch1"), ("2", "fail_percent_ch2"), ("3", "fail_percent_ch3")):
        assert sq.channel_plot_data(int(c), cd)[1] == want[k]
    assert cd["3"] == ()


@pytest.mark.parametrize("seed,tz", [(1, 0), (2, 3600), (3, -18000)])
def test_faststats_and_plot_data_equal_reference_shaped_layer(seed, tz):
    n_ch = 512 if seed == 1 else 11
    b = random_batch(seed, 900, 60, edge=False, n_channels=n_ch, hours=47, bad_meta=False)
    b["mux"][::17] = 0
    st = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
    cfg = dict(n_channels=n_ch, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=tz,
               ref_time_s=1_528_000_000, fail_q=7.0)
    t = po.build_tables(st, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg)
    ref_cd = _ref_shaped(b, st, n_ch, tz, 1_528_000_000)
    cd = sq.channel_dict_from_tables(t)
    want, got = po.FastStats_data(ref_cd), sq.faststats_data(cd)
    for k in ("FailSum", "PassSum", "table", "pie", "SmallestRead", "BiggestRead", "MeanReadLength", "TotalThroughput"):
        assert got[k] == want[k], k
    assert got["HoursDict"] == want["HoursDict"]
    for c in range(1, n_ch + 1):
        assert sq.channel_plot_data(c, cd) == po.ChannelPlotter_data(c, ref_cd)
        e, r = cd[str(c)], ref_cd[str(c)]
        assert (e == ()) == (r == ())
        if e:
            assert e[:3] == tuple(r[:3])
            assert {k: v[0] for k, v in e[4].items()} == {k: v[0] for k, v in r[4].items()}
    if n_ch == 512:
        assert sq.plate_data(cd) == po.result_plotting_data(ref_cd)
        assert sq.result_plotting(cd, write_html=False) == po.result_plotting_data(ref_cd)
    # the explicit histograms equal a direct count over the reference's raw vectors (plotly.js bins of :228-245)
    q = np.asarray(want["QualsVec"][1:], dtype=np.float64)        # [0] is the reference's sentinel
    assert got["q_hist"]["counts"] == [int(((q >= 6 + j) & (q < 7 + j)).sum()) for j in range(8)]
    g = np.asarray(want["GCVec"], dtype=np.float64)
    assert got["gc_hist"]["counts"] == [int(((g >= 5 * j) & ((g < 5 * j + 5) | (j == 19))).sum()) for j in range(20)]
    assert sum(got["len_hist"]["counts"]) == len(want["LengthsVec"]) - 1
    assert sq.gc_histogram(want["GCVec"]) == got["gc_hist"]["counts"]


def test_albacore_shape_and_summary_reader(tmp_path):
    rng = np.random.default_rng(5)
    hdr = "filename read_id run_id channel start_time duration num_events passes_filtering template_start " \
          "num_events_template template_duration num_called_template sequence_length_template mean_qscore_template " \
          "strand_score_template calibration_strand_genome_template a b c"
    lines = [hdr]
    for i in range(600):
        ch, start = int(rng.integers(1, 13)), round(float(rng.uniform(5.0, 47 * 3600)), 4)
        core = [str(ch), repr(start), "1.0", str(int(rng.integers(0, 9000))), "True", "0.0", "10",
                "%.3f" % rng.normal(9, 2)]
        lines.append(" ".join(["f", "r%d" % i, "u"] + core + ["p1", "p2", "p3", "p4", "p5"]))
    p = tmp_path / "sequencing_summary.txt"
    p.write_text("\n".join(lines) + "\n")
    rows = sq.summary_reader(str(p))
    ref_cd, ref_start = po.summary_reader(lines, n_channels=12)
    assert rows["ref_start"] == ref_start and len(rows["channel"]) == 600
    # the seconds handed to the GPU are localtime()'s floor of the fp64 difference (lib/seq_routines.py:846)
    sec = np.floor(rows["start"] - rows["ref_start"]).astype(np.int64)
    stats = dict(length=rows["length"], gc=np.zeros(600, np.int64), n=np.zeros(600, np.int64),
                 qsum=np.zeros(600, np.int64), err=[0] * 600, mean_q=rows["mean_q"], is_record=np.ones(600, np.uint8))
    cfg = dict(n_channels=12, n_mux=1, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0, ref_time_s=0)
    t = po.build_tables(stats, rows["channel"], np.ones(600, np.uint8), sec, None, cfg)
    cd = sq.channel_dict_from_tables(t, summary=True)
    for c in range(1, 13):
        e, r = cd[str(c)], ref_cd[str(c)]
        assert (e == ()) == (r == ())
        if e:
            assert e[:3] == r[:3] and e[4] == r[4]
        assert sq.channel_plot_data(c, cd) == po.ChannelPlotter_data(c, ref_cd, summary_flag=True)
    gct = [41.0, 0.0, 52.5, 100.0]
    want, got = po.FastStats_data(ref_cd, GCTable=gct), sq.faststats_data(cd, GCTable=gct)
    for k in ("FailSum", "PassSum", "table", "pie", "HoursDict"):
        assert got[k] == want[k], k
    assert got["gc_hist"]["counts"][8] == 1 and got["gc_hist"]["counts"][10] == 1 and got["gc_hist"]["counts"][19] == 1
    assert sq.gc_histogram((np.array([41, 0, 21, 5]), np.array([100, 7, 40, 5]))) == sq.gc_histogram(gct)


def test_cli_surface_matches_reference(tmp_path, monkeypatch):
    p = seqstats.build_parser(str(tmp_path))
    a = p.parse_args(["-i", str(tmp_path), "-l", "lab"])
    assert (a.fastq, a.albacore_summary, a.multi_read_fast5, a.output_dir, a.threads_number, a.verbose) == \
        (["no"], None, ["no"], None, 1, 1)                       # lib/seqstats.py:103-116
    a = p.parse_args(["-i", str(tmp_path), "-l", "x", "-fq", "yes", "-a", "s.txt", "-m", "y", "-o", "out", "-n", "8", "-v"])
    assert a.fastq == ["yes"] and a.albacore_summary == ["s.txt"] and a.output_dir == ["out"] and a.verbose == 1
    with pytest.raises(SystemExit):
        p.parse_args(["-i", str(tmp_path), "-l", "x", "-fq", "maybe"])
    with pytest.raises(SystemExit) as e:
        p.parse_args(["-i", str(tmp_path / "nope"), "-l", "x"])      # readable_dir, lib/seqstats.py:10-20
    assert e.value.code == 1
    seen = {}
    monkeypatch.setattr(sq, "run", lambda ags: seen.setdefault("ags", ags))
    monkeypatch.chdir(tmp_path)
    seqstats.ArgsReader(p.parse_args(["-i", str(tmp_path), "-l", "lab", "-a", "s.txt", "-n", "3"]))
    assert seen["ags"][:10] == [str(tmp_path), str(tmp_path), "lab", "PyPore_results", "1", "False", "3", "True",
                                "['s.txt']", "False"]              # lib/seqstats.py:50
    with pytest.raises(SystemExit):
        seqstats.run(["-i", str(tmp_path)])                          # missing -l, lib/seqstats.py:120-123
    from pypore_amd import pypore
    with pytest.raises(SystemExit):
        pypore.main(["pypore", "alignment"])
    with pytest.raises(SystemExit):
        pypore.main(["pypore"])


def _fq(title, seq, qual):
    return "@%s\n%s\n+\n%s\n" % (title, seq, qual)


def test_single_read_fast5_loader_on_stand_in_files(tmp_path):
    d0 = tmp_path / "pass" / "0"
    df = tmp_path / "fail" / "0"
    d0.mkdir(parents=True)
    df.mkdir(parents=True)
    now = time.time()
    specs = [("r_read_10_ch_100_strand.fast5", d0, 10, 100, 2, "2018-06-03T01:00:05Z", "2018-06-03T00:00:00Z", _fq("id10 x", "ACGT", "IIII")),
             ("r_read_11_ch_7_strand.fast5", d0, 11, 7, 4, "2018-06-03T02:30:00Z", "2018-06-03T00:00:10Z", _fq("id11", "GGCCN", "!!!!!")),
             ("r_read_12_ch_7_strand.fast5", df, 12, 7, 1, "2018-06-03T03:00:00Z", "2018-06-02T23:59:50Z", None)]
    for k, (name, d, rn, ch, mux, stamp, exp, fq) in enumerate(specs):
        fake_h5.write_single(str(d / name), rn, ch, mux, stamp, exp, fq)
        os.utime(d / name, (now + k, now + k))
    (d0 / ".hidden").write_text("x")
    broken = d0 / "r_read_13_ch_9_strand.fast5"
    broken.write_text(json.dumps({"attrs": {}, "datasets": {}}))
    os.utime(broken, (now + 9, now + 9))
    files, ref = f5r.dir_surfer(str(tmp_path), h5_open=fake_h5.FakeH5)
    assert sorted(os.path.basename(f) for f in files) == sorted([s[0] for s in specs] + [broken.name])
    # RefTime: min exp_start_time over the FIRST file (by mtime) of each leaf directory (lib/seq_routines.py:882-893)
    assert ref == min(f5r.mktime_local("2018-06-03T00:00:00Z"), f5r.mktime_local("2018-06-02T23:59:50Z"))
    batches = list(f5r.load_batches(files, 512, threads=2, h5_open=fake_h5.FakeH5))
    assert len(batches) == 1
    b = batches[0]
    assert b["n_reads"] == 3                                        # the broken file is skipped (except: del ds)
    order = np.argsort(b["start_time"])
    assert b["channel"][order].tolist() == [100, 7, 7] and b["mux"][order].tolist() == [2, 4, 1]
    assert b["fail_flag"][order].tolist() == [0, 0, 1]
    assert b["start_time"][order].tolist() == [int(f5r.mktime_local(s[5])) for s in specs]
    seqs = {b["bases"][int(b["offsets"][i]):int(b["offsets"][i + 1])].tobytes() for i in range(3)}
    assert seqs == {b"ACGT", b"GGCCN", b""}
    assert sorted(b["titles"]) == [b"", b"id10 x", b"id11"]
    with pytest.raises(RuntimeError, match="h5py"):
        f5r.get_content(str(d0 / specs[0][0]))                       # no h5py in this image: loud, no fallback
    with pytest.raises((ValueError, IndexError)):
        f5r.channel_of("nochannel.fast5")                            # the reference dies the same way (:909-910)


def test_multi_read_fast5_loader_on_stand_in_files(tmp_path):
    (tmp_path / "pass").mkdir()
    (tmp_path / "fail").mkdir()
    fake_h5.write_multi(str(tmp_path / "pass" / "a.fast5"),
                        [("read_a", 5, 1, "2018-06-03T00:10:00Z", _fq("a", "ACGTAC", "IIIIII")),
                         ("read_b", 6, 3, "2018-06-03T05:00:00Z", _fq("b", "GG", "++")),
                         ("read_c", 6, 3, "2018-06-03T05:00:00Z", "@c\nGGG\n+\n++\n")], "2018-06-03T00:00:00Z")
    fake_h5.write_multi(str(tmp_path / "fail" / "b.fast5"),
                        [("read_d", 9, 2, "2018-06-03T06:00:00Z", None)], "2018-06-03T00:00:00Z")
    files = mf5r.dir_surfer(str(tmp_path))
    assert len(files) == 2
    batches, ref = mf5r.load_batches(files, threads=2, h5_open=fake_h5.FakeH5)
    assert ref == f5r.mktime_local("2018-06-03T00:00:00Z")
    by = {os.path.basename(f): b for f, b in zip(files, batches)}
    a, f = by["a.fast5"], by["b.fast5"]
    assert a["n_reads"] == 2 and a["channel"].tolist() == [5, 6] and a["mux"].tolist() == [1, 3]   # read_c: bad lengths
    assert a["bases"].tobytes() == b"ACGTACGG" and a["offsets"].tolist() == [0, 6, 8]
    assert f["n_reads"] == 1 and f["fail_flag"].tolist() == [1] and f["offsets"].tolist() == [0, 0]


def test_bgzf_writer_and_pass_filter(tmp_path):
    chan = np.array([3, 1, 2, 1, 3], np.uint32)
    flags = np.array([0, 0, 1, 0, 2], np.uint8)
    assert fastq_out.passing_order(chan, flags).tolist() == [1, 3, 0]      # by channel, arrival order inside it
    big = b"ACGT" * 50000
    recs = [(b"r%d ch=%d" % (i, int(chan[i])), big[: 1000 * (i + 1) * 40], b"I" * (1000 * (i + 1) * 40)) for i in range(5)]
    batch = dict(n_reads=5, titles=[r[0] for r in recs], channel=chan,
                 bases=np.frombuffer(b"".join(r[1] for r in recs), np.uint8),
                 quals=np.frombuffer(b"".join(r[2] for r in recs), np.uint8),
                 offsets=np.concatenate(([0], np.cumsum([len(r[1]) for r in recs]))).astype(np.uint64))
    out = tmp_path / "res" / "x.fastq.gz"
    assert fastq_out.write_passing_batches([batch], dict(flags=flags), str(out)) == 3
    raw = out.read_bytes()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC" and raw.endswith(fastq_out._EOF)
    text = gzip.decompress(raw)
    assert text == b"".join(b"@" + recs[i][0] + b"\n" + recs[i][1] + b"\n+\n" + recs[i][2] + b"\n" for i in (1, 3, 0))
    # every BGZF block declares its own size and holds at most 64 KiB
    pos, blocks = 0, 0
    while pos < len(raw):
        bsize = int.from_bytes(raw[pos + 16:pos + 18], "little") + 1
        assert raw[pos:pos + 4] == b"\x1f\x8b\x08\x04" and bsize <= 65536
        pos += bsize
        blocks += 1
    assert pos == len(raw) and blocks > 3


def test_html_reports_are_written_with_the_reference_trace_names(tmp_path, monkeypatch):
    b = random_batch(4, 400, 80, edge=False, n_channels=512, hours=47, bad_meta=False)
    st = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
    cfg = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0,
               ref_time_s=1_528_000_000, fail_q=7.0)
    cd = sq.channel_dict_from_tables(po.build_tables(st, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg))
    monkeypatch.setattr(sq, "work_dir", str(tmp_path))
    monkeypatch.setattr(sq, "prefix", "lab")
    monkeypatch.setattr(sq, "out_dir", "res")
    d = sq.FastStats(cd)
    sq.result_plotting(cd)
    s = (tmp_path / "res" / "lab_sequencing_summary.html").read_text()
    for name in ("ReadsPerHour", "BasesPerHour", "MeanReadLengthPerHour", "READS TOTAL", "plotly_hover"):
        assert name in s
    assert d["html"].endswith("lab_sequencing_summary.html")
    s = (tmp_path / "res" / "lab_pore_activity_map.html").read_text()
    for name in ("ReadNHeat", "BaseNHeat", "Pore1", "Pore4", "% Pass", "plotly_click", "CHANNEL:"):
        assert name in s
    assert not (tmp_path / "lab_sequencing_summary.html").exists()       # moved, lib/seq_routines.py:92, 316


def test_cli_surface_matches_the_reference_option_table():
    """The option table of the reference's lib/seqstats.py:86-116, extracted as data from its source text into
    tests/golden/reference_constants.json (make_reference_constants.py): every reference option exists here with the
    same flags, default, choices, type and nargs; the default output directory is the reference's."""
    import json
    import os
    from pypore_amd import seqstats
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_constants.json")))
    parser = seqstats.build_parser("<tmpdir>")
    by_long = {}
    for a in parser._actions:
        for s in a.option_strings:
            if s.startswith("--"):
                by_long[s] = a
    assert len(ref["seqstats_cli"]) == 9
    for o in ref["seqstats_cli"]:
        a = by_long[o["long"]]
        assert o["short"] in a.option_strings, o
        if o["long"] == "--help":
            continue
        assert a.default == o["default"], (o, a.default)
        assert (None if a.choices is None else list(a.choices)) == o["choices"], (o, a.choices)
        assert (a.type is int) == (o["type"] == "int"), o
        assert a.nargs == o["nargs"], (o, a.nargs)
    import inspect
    assert "'%s'" % ref["default_output_dir"] in inspect.getsource(seqstats.ArgsReader)
    # output file names: <label>_sequencing_summary.html, <label>_pore_activity_map.html, <label>.fastq.gz
    from pypore_amd import seq_routines
    src = inspect.getsource(seq_routines)
    assert ref["output_suffixes"] == [".fastq.gz", "_pore_activity_map.html", "_sequencing_summary.html"]
    for suffix in ref["output_suffixes"]:
        assert "prefix + '%s'" % suffix in src, suffix
