"""`pypore seqstats` end to end on the GPU (CLI -> loaders -> C ABI -> tables -> summary), every mode, against the
oracle's REFERENCE-SHAPED layer (string tuples -> parsing_func / summary_reader -> FastStats_data), i.e. against the
restated reference pipeline rather than against the table layer the kernels mirror."""
import gzip
import os
import time

import numpy as np
import pytest

from oracle import pypore_oracle as po
import fake_h5

pytestmark = pytest.mark.gpu

ALPH = np.frombuffer(b"ACGTACGTNacgtSs", np.uint8)


def _records(seed, n, n_ch=512, hours=47):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.choice([0, 1, 5, 200, 3000, int(rng.integers(1, 20000))]))
        seq = bytes(rng.choice(ALPH, L))
        mu = rng.choice([4.0, 7.0, 12.0, 20.0])
        qual = bytes((33 + np.clip(np.rint(rng.normal(mu, 3, L)), 0, 60)).astype(np.uint8))
        ch = int(rng.integers(1, n_ch + 1))
        sec = int(rng.integers(0, hours * 3600))
        out.append(dict(i=i, seq=seq, qual=qual, ch=ch, sec=sec, mux=int(rng.integers(1, 5))))
    return out


def _iso(base, sec):
    return time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(base + sec))


def _fastq_text(recs, base):
    return b"".join(b"@read%d runid=r read=%d ch=%d start_time=%s\n%s\n+\n%s\n" % (
        r["i"], r["i"], r["ch"], _iso(base, r["sec"]).encode(), r["seq"], r["qual"]) for r in recs)


def _check_stats(got, want):
    for k in ("FailSum", "PassSum", "table", "pie", "HoursDict", "SmallestRead", "BiggestRead", "MeanReadLength",
              "TotalThroughput"):
        assert got[k] == want[k], (k, got[k], want[k])


def _plot_data(html_path):
    """The traces plotly wrote into an HTML file: the first JSON array behind ``Plotly.newPlot(``."""
    import json
    text = open(html_path).read()
    k = text.rindex("Plotly.newPlot(")       # the page also carries plotly.js itself, which has the words too
    k = text.index("[", k)
    data, _ = json.JSONDecoder().raw_decode(text[k:])
    return data, text


def _check_html_contract(res_dir, label, stats, per, n_channels=512):
    """What the reference's injected JavaScript and its readers rely on (lib/seq_routines.py:84, 207, 306-308): the trace
    names, the number of per-channel traces, and the summary table's cells."""
    data, text = _plot_data(os.path.join(res_dir, label + "_sequencing_summary.html"))
    names = [t.get("name") for t in data]
    for n in ("ReadsPerHour", "BasesPerHour", "MeanReadLengthPerHour"):
        assert names.count(n) == 1, (n, names)
        assert "divname == '%s'" % n in text                     # the hover glue keys on exactly these names
    tr = {t["name"]: t for t in data if t.get("name") in ("ReadsPerHour", "BasesPerHour", "MeanReadLengthPerHour")}
    assert [int(v) for v in tr["ReadsPerHour"]["y"]] == [v[0] for v in stats["HoursDict"]]
    assert [int(v) for v in tr["BasesPerHour"]["y"]] == [v[1] for v in stats["HoursDict"]]
    assert [float(v) for v in tr["MeanReadLengthPerHour"]["y"]] == [float(v[2]) for v in stats["HoursDict"]]
    tables = [t for t in data if t.get("type") == "table"]
    assert len(tables) == 1
    cells = [c[0] if isinstance(c, list) else c for c in tables[0]["cells"]["values"]]
    num = lambda v: None if v is None else float(v)
    assert [num(c) for c in cells] == [num(v) for v in stats["table"]], (cells, stats["table"])
    data2, text2 = _plot_data(os.path.join(res_dir, label + "_pore_activity_map.html"))
    names2 = [t.get("name") for t in data2]
    assert names2[:2] == ["ReadNHeat", "BaseNHeat"] and data2[0]["type"] == "heatmap"
    assert len(data2) == 2 + per * n_channels
    if per == 10:
        for k in range(1, 5):
            assert names2.count("Pore%d" % k) == 2 * n_channels
            assert "divname == 'Pore%d'" % k in text2
    else:
        assert names2.count("Channel") == 2 * n_channels and "divname == 'Channel'" in text2
    assert names2.count("Fail") == n_channels and names2.count("Pass") == n_channels
    return data, data2


class _PinnedWatch:
    """qc.SUBMIT_HOOK: every host batch handed to ppqc_submit / ppqc_submit_fastq, and whether all of its arrays lie in
    page-locked staging slots (north_star: loaders pack into pinned buffers)."""

    def __init__(self):
        self.calls = []

    def __call__(self, kind, arrays):
        from pypore_amd import staging
        self.calls.append((kind, all(staging.in_pinned_slot(a, n) for a, n in arrays.values() if n)))


def _mean_q(qual):
    return po.mean_q_from_fixed(po.err_sum_fixed(qual), len(qual)) if qual else 0.0


def test_albacore_mode_end_to_end(tmp_path, monkeypatch):
    from pypore_amd import seqstats, seq_routines as sq
    recs = _records(1, 1500)
    (tmp_path / "fq" / "pass").mkdir(parents=True)
    (tmp_path / "fq" / "fail").mkdir(parents=True)
    base = 1_528_000_000
    texts = [_fastq_text(recs[:900], base), _fastq_text(recs[900:], base)]
    (tmp_path / "fq" / "pass" / "a.fastq").write_bytes(texts[0])
    (tmp_path / "fq" / "fail" / "b.fastq").write_bytes(texts[1])
    hdr = "filename read_id run_id channel start_time duration num_events passes_filtering template_start " \
          "num_events_template template_duration num_called_template sequence_length_template mean_qscore_template " \
          "strand_score_template calibration_strand_genome_template a b c"
    lines = [hdr]
    rng = np.random.default_rng(2)
    for r in recs:
        core = [str(r["ch"]), repr(r["sec"] + round(float(rng.random()), 4) + 3.25), "1.0", str(len(r["seq"])), "True",
                "0.0", "10", "%.3f" % _mean_q(r["qual"])]
        lines.append(" ".join(["f", "read%d" % r["i"], "u"] + core + ["p1", "p2", "p3", "p4", "p5"]))
    table = tmp_path / "sequencing_summary.txt"
    table.write_text("\n".join(lines) + "\n")
    monkeypatch.chdir(tmp_path)
    from pypore_amd import qc
    watch = _PinnedWatch()
    monkeypatch.setattr(qc, "SUBMIT_HOOK", watch)
    cd, stats, plates = seqstats.run(["-i", str(tmp_path / "fq"), "-l", "alb", "-a", str(table), "-n", "3", "-o", "res"])
    assert [c for c in watch.calls] == [("fastq", True), ("fastq", True)]      # both files: text read into pinned slots
    ref_cd, _ = po.summary_reader(lines, n_channels=512, tz_offset_s=sq.local_tz_offset())
    gc_list = po.fastq_reader(texts[0]) + po.fastq_reader(texts[1])     # order is irrelevant to every consumer
    want = po.FastStats_data(ref_cd, GCTable=gc_list)
    _check_stats(stats, want)
    for c in range(1, 513):
        e, r = cd[str(c)], ref_cd[str(c)]
        assert (e == ()) == (r == ())
        if e:
            assert e[:3] == r[:3] and e[4] == r[4]
    assert plates == po.result_plotting_data(ref_cd)
    g = np.asarray(want["GCVec"])
    assert stats["gc_hist"]["counts"] == [int(((g >= 5 * j) & ((g < 5 * j + 5) | (j == 19))).sum()) for j in range(20)]
    q = np.asarray(want["QualsVec"][1:])
    assert stats["q_hist"]["counts"] == [int(((q >= 6 + j) & (q < 7 + j)).sum()) for j in range(8)]
    assert cd.seq_tables["n50"] == po.n50([len(r["seq"]) for r in recs])
    for f in ("alb_sequencing_summary.html", "alb_pore_activity_map.html"):
        assert (tmp_path / "res" / f).stat().st_size > 10000
    _check_html_contract(str(tmp_path / "res"), "alb", stats, per=4)


def _ref_shaped(recs, ref, tz, n_ch=512, times=None, multi_read_single_quirk=False):
    tuples = []
    for k, r in enumerate(recs):
        t = times[k] if times is not None else ref + r["sec"]
        tuples.append(po.get_content_tuple(r["ch"], r["mux"], t, len(r["seq"]), _mean_q(r["qual"]), po.GC(r["seq"]),
                                           fail=bool(r.get("fail"))))
    return po.build_channel_dict(tuples, float(ref), n_channels=n_ch, tz_offset_s=tz,
                                 multi_read_single_quirk=multi_read_single_quirk)


def test_fastq_only_mode_with_fastq_output(tmp_path, monkeypatch):
    from pypore_amd import seqstats
    recs = _records(3, 1200)
    for r in recs:
        r["mux"] = 1                       # FASTQ headers carry no mux
    base = 1_528_000_000
    (tmp_path / "in").mkdir()
    (tmp_path / "in" / "a.fastq").write_bytes(_fastq_text(recs[:700], base))
    with gzip.open(tmp_path / "in" / "b.fastq.gz", "wb") as fh:
        fh.write(_fastq_text(recs[700:], base))
    monkeypatch.chdir(tmp_path)
    cd, stats, plates = seqstats.run(["-i", str(tmp_path / "in"), "-l", "fqo", "-f", "fastq", "-fq", "yes"])
    ref = base + min(r["sec"] for r in recs)
    ref_cd = _ref_shaped(recs, ref, 0, times=[base + r["sec"] for r in recs])
    _check_stats(stats, po.FastStats_data(ref_cd))
    assert plates == po.result_plotting_data(ref_cd)
    assert stats["n50"] == po.n50([len(r["seq"]) for r in recs])
    # <label>.fastq.gz: reads with mean_q >= 7.0, by channel, arrival order inside a channel (lib/fast5_reader.py:112-114)
    keep = [r for r in recs if _mean_q(r["qual"]) >= 7.0]
    keep.sort(key=lambda r: r["ch"])
    want = b"".join(b"@read%d runid=r read=%d ch=%d start_time=%s\n%s\n+\n%s\n" % (
        r["i"], r["i"], r["ch"], _iso(base, r["sec"]).encode(), r["seq"], r["qual"]) for r in keep)
    assert gzip.decompress((tmp_path / "PyPore_results" / "fqo.fastq.gz").read_bytes()) == want
    # without -fq the text is tokenised on the GPU; a file with blank lines between records (which SeqIO tolerates
    # and the strict GPU tokenizer refuses) takes the host packer: same summary either way
    t0 = _fastq_text(recs[:700], base)
    t = t0.replace(b"\n@read5 ", b"\n\n@read5 ", 1)
    assert len(t) == len(t0) + 1
    (tmp_path / "in" / "a.fastq").write_bytes(t)
    cd2, stats2, plates2 = seqstats.run(["-i", str(tmp_path / "in"), "-l", "fqg", "-f", "fastq", "-o", "res_gpu"])
    _check_stats(stats2, po.FastStats_data(ref_cd))
    assert plates2 == plates and dict(cd2) == dict(cd) and stats2["n50"] == stats["n50"]
    assert not (tmp_path / "res_gpu" / "fqg.fastq.gz").exists()
    # a record without ch= is an error (the reference dies in int(), lib/seq_routines.py:51)
    (tmp_path / "in" / "c.fastq").write_bytes(b"@x start_time=2018-06-03T00:00:00Z\nACGT\n+\nIIII\n")
    with pytest.raises(ValueError, match="ch="):
        seqstats.run(["-i", str(tmp_path / "in"), "-l", "bad", "-f", "fastq"])


def test_fast5_modes_on_stand_in_hdf5(tmp_path, monkeypatch):
    from pypore_amd import seq_routines as sq, fast5_reader as f5r
    monkeypatch.setattr(sq, "H5_OPEN", fake_h5.FakeH5)
    monkeypatch.chdir(tmp_path)
    recs = _records(5, 300, hours=40)
    base = 1_528_000_000
    exp = _iso(base, -20)
    # single-read: one file per read, channel in the file name, fail status in the path
    for sub in ("s/pass/0", "s/fail/0"):
        (tmp_path / sub).mkdir(parents=True)
    for r in recs:
        r["fail"] = r["i"] % 7 == 0
        d = tmp_path / ("s/fail/0" if r["fail"] else "s/pass/0")
        fq = None if r["fail"] else "@read%d\n%s\n+\n%s\n" % (r["i"], r["seq"].decode(), r["qual"].decode())
        # albacore's own summary attributes (lib/fast5_reader.py:55-56): consistent with the record, but for two reads --
        # one whose sequence_length is off by one, one whose mean_qscore lies on the other side of 7.0 -- and absent on some
        mq = _mean_q(r["qual"])
        r["summary"] = None if r["i"] % 5 == 0 else (len(r["seq"]) + (1 if r["i"] == 11 else 0),
                                                      (9.0 if mq < 7.0 else 6.5) if r["i"] == 12 else mq)
        fake_h5.write_single(str(d / ("x_read_%d_ch_%d_strand.fast5" % (r["i"], r["ch"]))), r["i"], r["ch"], r["mux"],
                             _iso(base, r["sec"]), exp, fq, r["summary"])
    assert not recs[11]["fail"] and not recs[12]["fail"]
    n_attr = sum(1 for r in recs if not r["fail"] and r["summary"] is not None)
    ref = f5r.mktime_local(exp)
    times = [int(f5r.mktime_local(_iso(base, r["sec"]))) for r in recs]
    ref_cd = _ref_shaped(recs, ref, sq.local_tz_offset(), times=times)
    from pypore_amd import qc
    watch = _PinnedWatch()
    monkeypatch.setattr(qc, "SUBMIT_HOOK", watch)
    cd, stats, plates = sq.run([str(tmp_path), str(tmp_path / "s"), "one", "res1", "1", "True", "2", "False", "None", "False"])
    assert watch.calls and all(c == ("csr", True) for c in watch.calls)       # CSR batches packed in pinned slots
    _check_stats(stats, po.FastStats_data(ref_cd))
    assert stats["albacore_check"] == dict(reads_with_attributes=n_attr, length_mismatch=1, fail_mismatch=1)
    _check_html_contract(str(tmp_path / "res1"), "one", stats, per=10)
    assert plates == po.result_plotting_data(ref_cd)
    for c in range(1, 513):
        assert sq.channel_plot_data(c, cd) == po.ChannelPlotter_data(c, ref_cd)
    out = gzip.decompress((tmp_path / "res1" / "one.fastq.gz").read_bytes())
    passing = {r["i"] for r in recs if not r["fail"] and _mean_q(r["qual"]) >= 7.0}
    assert out.count(b"\n+\n") == len(passing) and all(b"@read%d\n" % i in out for i in passing)
    # multi-read: the same reads in three files, fail status from the file path
    (tmp_path / "m" / "pass").mkdir(parents=True)
    (tmp_path / "m" / "fail").mkdir(parents=True)
    good = [r for r in recs if not r["fail"]]
    bad = [r for r in recs if r["fail"]]
    def entry(r, with_fq):
        fq = "@read%d\n%s\n+\n%s\n" % (r["i"], r["seq"].decode(), r["qual"].decode()) if with_fq else None
        return ("read_%04d" % r["i"], r["ch"], r["mux"], _iso(base, r["sec"]), fq, r["summary"] if with_fq else None)
    fake_h5.write_multi(str(tmp_path / "m" / "pass" / "a.fast5"), [entry(r, True) for r in good[:100]], exp)
    fake_h5.write_multi(str(tmp_path / "m" / "pass" / "b.fast5"), [entry(r, True) for r in good[100:]], exp)
    fake_h5.write_multi(str(tmp_path / "m" / "fail" / "c.fast5"), [entry(r, False) for r in bad], exp)
    cd2, stats2, plates2 = sq.run([str(tmp_path), str(tmp_path / "m"), "multi", "res2", "1", "False", "2", "False",
                                   "None", "True"])
    # multi-read mode goes through channel_parser, which takes the read of a single-read channel under EVERY mux
    # (lib/multi_read_fast5_reader.py:64-67): the expectation is built with that branch, and the fixture has such
    # channels, failing and passing ones (300 reads over 512 channels)
    ref_cd_m = _ref_shaped(recs, ref, sq.local_tz_offset(), times=times, multi_read_single_quirk=True)
    singles = [c for c in range(1, 513) if ref_cd_m[str(c)] and ref_cd_m[str(c)][0] == 1]
    assert len(singles) > 50 and any(ref_cd_m[str(c)][2] == 4 for c in singles) and \
        any(ref_cd_m[str(c)][2] == 0 for c in singles)
    assert po.FastStats_data(ref_cd_m)["table"] != po.FastStats_data(ref_cd)["table"]     # the branch matters here
    _check_stats(stats2, po.FastStats_data(ref_cd_m))
    assert stats2["albacore_check"] == dict(reads_with_attributes=n_attr, length_mismatch=1, fail_mismatch=1)
    assert plates2 == plates
    for c in range(1, 513):
        e, r = cd2[str(c)], ref_cd_m[str(c)]
        assert (e == ()) == (r == ())
        if e:
            assert tuple(e[:3]) == tuple(r[:3]), (c, e[:3], r[:3])
        assert sq.channel_plot_data(c, cd2) == po.ChannelPlotter_data(c, ref_cd_m)


def _run_cli(args, cwd, extra_env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "tests"), env.get("PYTHONPATH", "")])
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-m", "pypore_amd.pypore", "seqstats"] + args, cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=280)
    return r


def test_cli_gpus_2_equals_one_process(tmp_path):
    """`pypore seqstats --gpus 2` started plainly: the CLI starts its two rank processes itself (torchrun child), every
    rank takes a contiguous share of the files, the tables are merged by finalize_distributed (time_mode 1 across ranks:
    every channel's reads are on both) and rank 0 writes the outputs.  Two ranks on the ONE GPU of the test box, so over
    gloo (PPQC_DIST_BACKEND; RCCL refuses two ranks on a device).  Everything written must equal the one-process run:
    every trace of both HTML files, and the decompressed <label>.fastq.gz byte for byte."""
    base = 1_528_000_000
    recs = _records(11, 900)
    for r in recs:
        r["mux"] = 1
    (tmp_path / "in").mkdir()
    cuts = [0, 200, 450, 460, 900]
    for k in range(4):
        (tmp_path / "in" / ("f%d.fastq" % k)).write_bytes(_fastq_text(recs[cuts[k]:cuts[k + 1]], base))
    gloo = {"PPQC_DIST_BACKEND": "gloo"}
    for label, extra in (("g1", []), ("g2", ["--gpus", "2"])):
        r = _run_cli(["-i", str(tmp_path / "in"), "-l", label, "-f", "fastq", "-fq", "yes", "-o", "res_" + label] + extra,
                     str(tmp_path), gloo)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    a = gzip.decompress((tmp_path / "res_g1" / "g1.fastq.gz").read_bytes())
    b = gzip.decompress((tmp_path / "res_g2" / "g2.fastq.gz").read_bytes())
    assert a == b and a.count(b"\n+\n") > 300
    assert not [f for f in os.listdir(tmp_path / "res_g2") if ".part" in f]
    for f in ("_sequencing_summary.html", "_pore_activity_map.html"):
        d1, _ = _plot_data(str(tmp_path / "res_g1" / ("g1" + f)))
        d2, _ = _plot_data(str(tmp_path / "res_g2" / ("g2" + f)))
        assert d1 == d2, f
    # albacore mode: rows of the table and FASTQ files both shared out over the ranks
    (tmp_path / "fq" / "pass").mkdir(parents=True)
    for k in range(3):
        (tmp_path / "fq" / "pass" / ("p%d.fastq" % k)).write_bytes(_fastq_text(recs[300 * k:300 * (k + 1)], base))
    hdr = "filename read_id run_id channel start_time duration num_events passes_filtering template_start " \
          "num_events_template template_duration num_called_template sequence_length_template mean_qscore_template " \
          "strand_score_template calibration_strand_genome_template a b c"
    lines = [hdr]
    for r in recs:
        core = [str(r["ch"]), repr(r["sec"] + 0.5), "1.0", str(len(r["seq"])), "True", "0.0", "10", "%.3f" % _mean_q(r["qual"])]
        lines.append(" ".join(["f", "read%d" % r["i"], "u"] + core + ["p1", "p2", "p3", "p4", "p5"]))
    (tmp_path / "summary.txt").write_text("\n".join(lines) + "\n")
    for label, extra in (("a1", []), ("a2", ["--gpus", "2"])):
        r = _run_cli(["-i", str(tmp_path / "fq"), "-l", label, "-a", str(tmp_path / "summary.txt"), "-o", "res_" + label] + extra,
                     str(tmp_path), gloo)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    for f in ("_sequencing_summary.html", "_pore_activity_map.html"):
        d1, _ = _plot_data(str(tmp_path / "res_a1" / ("a1" + f)))
        d2, _ = _plot_data(str(tmp_path / "res_a2" / ("a2" + f)))
        assert d1 == d2, f
    # single-read Fast5 through the stand-in HDF5 (PYPORE_H5_OPENER reaches the rank processes), with -fq
    exp = _iso(base, -20)
    for sub in ("s/pass/0", "s/fail/0"):
        (tmp_path / sub).mkdir(parents=True)
    for r in _records(12, 240, hours=40):
        fail = r["i"] % 7 == 0
        d = tmp_path / ("s/fail/0" if fail else "s/pass/0")
        fq = None if fail else "@read%d\n%s\n+\n%s\n" % (r["i"], r["seq"].decode(), r["qual"].decode())
        fake_h5.write_single(str(d / ("x_read_%d_ch_%d_strand.fast5" % (r["i"], r["ch"]))), r["i"], r["ch"], r["mux"],
                             _iso(base, r["sec"]), exp, fq)
    h5 = dict(gloo, PYPORE_H5_OPENER="fake_h5:FakeH5")
    for label, extra in (("s1", []), ("s2", ["--gpus", "2"])):
        r = _run_cli(["-i", str(tmp_path / "s"), "-l", label, "-fq", "yes", "-n", "2", "-o", "res_" + label] + extra,
                     str(tmp_path), h5)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    for f in ("_sequencing_summary.html", "_pore_activity_map.html"):
        d1, _ = _plot_data(str(tmp_path / "res_s1" / ("s1" + f)))
        d2, _ = _plot_data(str(tmp_path / "res_s2" / ("s2" + f)))
        assert d1 == d2, f
    assert gzip.decompress((tmp_path / "res_s1" / "s1.fastq.gz").read_bytes()) == \
        gzip.decompress((tmp_path / "res_s2" / "s2.fastq.gz").read_bytes())
    # multi-read Fast5: channel_parser's single-read-channel rule (lib/multi_read_fast5_reader.py:64-67) looks at a channel's
    # read count in the whole job -- the ranks add their counts up, so two ranks give the tables of one, and both those of
    # parsing_func(multi_read_single_quirk=True).  240 reads over 512 channels in four files: single-read channels on
    # either rank, and channels with one read on each
    mrecs = _records(13, 240, hours=40)
    (tmp_path / "m" / "pass").mkdir(parents=True)
    (tmp_path / "m" / "fail").mkdir(parents=True)
    for r in mrecs:
        r["fail"] = r["i"] % 7 == 0
    good = [r for r in mrecs if not r["fail"]]
    bad = [r for r in mrecs if r["fail"]]

    def entry(r, with_fq):
        fq = "@read%d\n%s\n+\n%s\n" % (r["i"], r["seq"].decode(), r["qual"].decode()) if with_fq else None
        return ("read_%04d" % r["i"], r["ch"], r["mux"], _iso(base, r["sec"]), fq)
    for name, part in (("pass/a", good[:60]), ("pass/b", good[60:130]), ("pass/c", good[130:])):
        fake_h5.write_multi(str(tmp_path / "m" / (name + ".fast5")), [entry(r, True) for r in part], exp)
    fake_h5.write_multi(str(tmp_path / "m" / "fail" / "d.fast5"), [entry(r, False) for r in bad], exp)
    for label, extra in (("m1", []), ("m2", ["--gpus", "2"])):
        r = _run_cli(["-i", str(tmp_path / "m"), "-l", label, "-m", "yes", "-n", "2", "-o", "res_" + label] + extra,
                     str(tmp_path), h5)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        assert "is not applied" not in r.stdout + r.stderr
    from pypore_amd import seq_routines as sq, fast5_reader as f5r
    ref_cd_m = _ref_shaped(mrecs, f5r.mktime_local(exp), sq.local_tz_offset(),
                           times=[int(f5r.mktime_local(_iso(base, r["sec"]))) for r in mrecs], multi_read_single_quirk=True)
    assert sum(1 for c in range(1, 513) if ref_cd_m[str(c)] and ref_cd_m[str(c)][0] == 1) > 50
    want_m = po.FastStats_data(ref_cd_m)
    for f in ("_sequencing_summary.html", "_pore_activity_map.html"):
        d1, _ = _plot_data(str(tmp_path / "res_m1" / ("m1" + f)))
        d2, _ = _plot_data(str(tmp_path / "res_m2" / ("m2" + f)))
        assert d1 == d2, f
        if f.startswith("_seq"):
            tr = {t["name"]: t for t in d2 if t.get("name") in ("ReadsPerHour", "BasesPerHour")}
            assert [int(v) for v in tr["ReadsPerHour"]["y"]] == [v[0] for v in want_m["HoursDict"]]
            assert [int(v) for v in tr["BasesPerHour"]["y"]] == [v[1] for v in want_m["HoursDict"]]
            cells = [c[0] if isinstance(c, list) else c for c in [t for t in d2 if t.get("type") == "table"][0]["cells"]["values"]]
            assert [float(c) for c in cells] == [float(v) for v in want_m["table"]]
    # a rank count the launcher cannot give must fail loudly, not run on fewer ranks
    r = _run_cli(["-i", str(tmp_path / "in"), "-l", "bad", "-f", "fastq", "--gpus", "2"], str(tmp_path),
                 {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
