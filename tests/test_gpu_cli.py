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
    cd, stats, plates = seqstats.run(["-i", str(tmp_path / "fq"), "-l", "alb", "-a", str(table), "-n", "3", "-o", "res"])
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
        fake_h5.write_single(str(d / ("x_read_%d_ch_%d_strand.fast5" % (r["i"], r["ch"]))), r["i"], r["ch"], r["mux"],
                             _iso(base, r["sec"]), exp, fq)
    ref = f5r.mktime_local(exp)
    times = [int(f5r.mktime_local(_iso(base, r["sec"]))) for r in recs]
    ref_cd = _ref_shaped(recs, ref, sq.local_tz_offset(), times=times)
    cd, stats, plates = sq.run([str(tmp_path), str(tmp_path / "s"), "one", "res1", "1", "True", "2", "False", "None", "False"])
    _check_stats(stats, po.FastStats_data(ref_cd))
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
        return ("read_%04d" % r["i"], r["ch"], r["mux"], _iso(base, r["sec"]), fq)
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
    assert plates2 == plates
    for c in range(1, 513):
        e, r = cd2[str(c)], ref_cd_m[str(c)]
        assert (e == ()) == (r == ())
        if e:
            assert tuple(e[:3]) == tuple(r[:3]), (c, e[:3], r[:3])
        assert sq.channel_plot_data(c, cd2) == po.ChannelPlotter_data(c, ref_cd_m)
