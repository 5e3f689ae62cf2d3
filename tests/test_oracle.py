"""The oracle against the hand-derived known answers, its two layers against each other, and the C restatement
against the Python one.  (No reference fixture exists to pin it further: see DESIGN.md, 'parity unpinned'.)"""
import json
import os

import numpy as np
import pytest

from oracle import pypore_oracle as po
from util import random_batch, assert_tables_equal
from golden.make_golden import toy_batch, toy_reads, TOY_CFG, REF

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def test_gc_known_answers():
    for seq, want in KAT["gc_percent"]:
        assert po.GC(seq) == want
    assert po.GC(b"ACTGN") == 40.0


def test_hour_wrap_known_answers():
    for raw, want in KAT["hour_wrap"]:
        assert po._hour_wrap(raw) == want
    # deltas in hours -> local hour of day (tz = UTC) -> wrap fix
    for deltas, want in (([23, 25, 26], [23, 1, 2]), ([0, 23, 25, 26, 47, 49], [0, 23, 25, 26, 47, 25]), ([25, 26, 27], [1, 2, 3])):
        assert po._hour_wrap([po.local_hour(d * 3600.0) for d in deltas]) == want
    assert po.local_hour(-1.0) == 23 and po.local_hour(3599.9) == 0 and po.local_hour(0, 3600) == 1


def test_mean_q_n50_ncount_known_answers():
    for quals, ont, arith in KAT["mean_q"]:
        q = quals.encode()
        assert abs(po.mean_q_ont(q) - ont) <= 1e-12 * max(1.0, ont)
        assert abs(po.mean_q_from_fixed(po.err_sum_fixed(q), len(q)) - ont) <= 1e-9 * max(1.0, ont)
        assert sum(po.phred_decode(q)) / len(q) == arith
    assert po.mean_q_ont(b"") == 0.0
    for lens, want in KAT["n50"]:
        assert po.n50(lens) == want
    for seq, want in KAT["n_count"]:
        assert po.n_count(seq.encode()) == want


def test_python2_semantics_helpers():
    for x, want in KAT["py2_round"]:
        assert po.py2_round(x) == want
    assert po.py2_round(82.29166, 1) == 82.3 and po.py2_round(2.675, 2) == 2.68 and po.py2_round(0.125, 2) == 0.13
    for x, want in KAT["py2_str_float"]:
        assert po.py2_str_float(x) == want


def test_fixed_point_threshold_is_consistent():
    # a read of constant quality q sits exactly on threshold q: pass at fail_q = q, fail just below
    for q in (0, 7, 20, 93):
        L = 1234
        S = po.err_sum_fixed(bytes([33 + q]) * L)
        assert S == L * po.q_threshold(float(q))
    assert po.err_lut()[33] == 1 << 52 and po.err_lut()[0] == 1 << 52 and po.err_lut()[255] == po.err_lut()[126]
    # fixed-point mean_q stays within 1e-9 of the float spec over the whole valid range
    rng = np.random.default_rng(0)
    for _ in range(50):
        q = rng.integers(33, 127, int(rng.integers(1, 3000))).astype(np.uint8).tobytes()
        a, b = po.mean_q_ont(q), po.mean_q_from_fixed(po.err_sum_fixed(q), len(q))
        assert abs(a - b) <= 1e-9 * max(abs(a), 1e-3)


def _toy_channel_dict():
    """The toy run through the REFERENCE-SHAPED layer: 6-string tuples -> parsing_func per channel."""
    tuples = []
    for ch, mux, hours, bases, quals, fail in toy_reads():
        st = po.per_read_stats_py(bases, quals)
        tuples.append(po.get_content_tuple(ch, mux, REF + int(hours * 3600), st["length"], st["mean_q"],
                                           po.GC(bases), fail=bool(fail)))
    return po.build_channel_dict(tuples, float(REF), n_channels=3)


def test_toy_run_reference_shaped_hand_derived():
    want = KAT["toy_run"]
    cd = _toy_channel_dict()
    assert cd["3"] == ()
    assert [cd[c][0] if cd[c] else 0 for c in "123"] == want["chan_reads"]
    assert [cd[c][1] if cd[c] else 0 for c in "123"] == want["chan_bases"]
    assert [cd[c][2] if cd[c] else 0 for c in "123"] == want["chan_fail"]
    for c, key in (("1", "mux_productivity_ch1"), ("2", "mux_productivity_ch2")):
        got = {k: {h: list(v) for h, v in cd[c][4][k][0].items()} for k in "1234"}
        assert got == want[key]
    # the fail-path read keeps the defaults of lib/fast5_reader.py:37-40
    assert ("25", "0", "6.99", "2", "0") in [tuple(str(x) for x in r) for r in cd["1"][3]]
    fs = po.FastStats_data(cd)
    assert fs["FailSum"] == want["FailSum"] and fs["PassSum"] == want["PassSum"]
    for h, (r, b, m) in want["hours"].items():
        assert fs["HoursDict"][int(h)] == [r, b, m]
    assert fs["table"] == want["table"] and fs["pie"] == want["pie"]
    assert fs["LengthsVec"][0] == 0 and len(fs["LengthsVec"]) == 10
    for c, k in (("1", "fail_percent_ch1"), ("2", "fail_percent_ch2"), ("3", "fail_percent_ch3")):
        assert po.ChannelPlotter_data(int(c), cd)[1] == want[k]


def test_toy_run_table_shaped_matches_reference_shaped_and_golden():
    b = toy_batch()
    st = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
    t = po.build_tables(st, b["channel"], b["mux"], b["time"], b["fail_flag"], TOY_CFG)
    want = KAT["toy_run"]
    assert t["chan_reads"].tolist() == want["chan_reads"] and t["chan_bases"].tolist() == want["chan_bases"]
    assert t["chan_fail"].tolist() == want["chan_fail"]
    assert t["hour"].tolist() == [0, 23, 25, 26, 47, 25, 23, 0, 1]
    # rebuild the reference's ChannelDict shape from the tables and run the reference-shaped FastStats on it
    cd = po.channel_dict_from_tables(t, TOY_CFG)
    ref_cd = _toy_channel_dict()
    for c in "123":
        if not ref_cd[c]:
            assert cd[c] == ()
            continue
        assert cd[c][:3] == ref_cd[c][:3]
        assert {k: v[0] for k, v in cd[c][4].items()} == {k: v[0] for k, v in ref_cd[c][4].items()}
    gold = json.load(open(os.path.join(HERE, "golden", "toy_run.json")))
    for k, v in gold["tables"].items():
        assert np.array_equal(np.asarray(t[k]).astype(np.int64), np.asarray(v, dtype=np.int64)), k
    assert [int(x) for x in st["err"]] == gold["per_read"]["err"]
    assert t["n50"] == 2000 and t["totals"][po.TOT_BAD_MUX] == 1


@pytest.mark.parametrize("seed,mode,tz,quirk", [(1, 1, 0, False), (2, 1, 3600, False), (3, 1, -18000, False),
                                                 (4, 1, 0, True), (5, 1, 7200, True)])
def test_table_layer_equals_reference_layer_on_random_runs(seed, mode, tz, quirk, c_oracle):
    """parsing_func / FastStats_data on string tuples == build_tables on arrays, including the wrap quirk and, with
    ``quirk``, channel_parser's single-read channel (lib/multi_read_fast5_reader.py:64-67 == compat_flags bit 0)."""
    n_ch = 9 if not quirk else 40
    b = random_batch(seed, 700 if not quirk else 60, 60, edge=False, n_channels=n_ch, hours=47, ref=1_528_000_000,
                     bad_meta=False)
    b["mux"][::17] = 0
    if quirk:      # single-read channels of every kind: mux 0, flagged as failed, low quality
        b["mux"][::5] = 0
        b["quals"][: int(b["offsets"][len(b["channel"]) // 2])] = 33 + 3
        counts = np.bincount(b["channel"], minlength=n_ch + 1)
        assert (counts == 1).sum() >= 5
    st = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
    cfg = dict(n_channels=n_ch, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=mode, tz_offset_s=tz,
               ref_time_s=1_528_000_000, fail_q=7.0, compat_flags=1 if quirk else 0)
    t = po.build_tables(st, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg)
    assert_tables_equal(c_oracle.tables(c_oracle.per_read(b["bases"], b["quals"], b["offsets"]), b["channel"], b["mux"],
                                        b["time"], b["fail_flag"], cfg), t, "C oracle vs Python oracle")
    tuples = []
    for i in range(len(b["channel"])):
        a, e = int(b["offsets"][i]), int(b["offsets"][i + 1])
        # the reference's float comparison `float(q) < 7.0` is made on the 12-digit string of mean_q
        tuples.append(po.get_content_tuple(int(b["channel"][i]), int(b["mux"][i]), int(b["time"][i]),
                                           int(st["length"][i]), float(st["mean_q"][i]),
                                           po.GC(b["bases"][a:e].tobytes()), fail=bool(b["fail_flag"][i])))
    cd = po.build_channel_dict(tuples, 1_528_000_000.0, n_channels=n_ch, tz_offset_s=tz, multi_read_single_quirk=quirk)
    for c in range(n_ch):
        e = cd[str(c + 1)]
        assert (e[0] if e else 0) == t["chan_reads"][c] and (e[1] if e else 0) == t["chan_bases"][c]
        assert (e[2] if e else 0) == t["chan_fail"][c]
        for m in range(4):
            d = e[4][str(m + 1)][0] if e else {}
            for h in range(48):
                r, bs = d.get(str(h), (0, 0))
                assert r == t["cmh_reads"][c, m, h] and bs == t["cmh_bases"][c, m, h]
    fs = po.FastStats_data(cd)
    assert [h[0] for h in fs["HoursDict"]] == t["hour_reads"].tolist()
    assert [h[1] for h in fs["HoursDict"]] == t["hour_bases"].tolist()
    assert fs["FailSum"] == int(t["chan_fail"].sum()) and fs["PassSum"] == int(t["hour_reads"].sum())
    if not quirk:
        assert fs["PassSum"] == int(t["totals"][po.TOT_N_BINNED])
    else:
        assert fs["PassSum"] > int(t["totals"][po.TOT_N_BINNED])     # single-read channels count once per mux
    assert fs["BiggestRead"] == t["max_len"]
    pos = [int(x) for x in list(t["chan_min_len"]) + list(t["chan_max_len"]) if 0 < int(x) < po.I64_MAX]
    assert fs["SmallestRead"] == min(pos)


def test_reads_without_a_start_time_have_no_hour_and_do_not_disturb_the_order(c_oracle):
    """INT64_MIN = no start time (a FASTQ header without start_time=): counted in TOT_NO_TIME, no hour bin, and the timed
    reads of the channel keep the hours they have without it, in both time modes."""
    b = random_batch(21, 400, 50, edge=False, n_channels=6, hours=47, bad_meta=False)
    st = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
    t2 = b["time"].copy()
    gone = np.arange(len(t2)) % 9 == 4
    t2[gone] = -(1 << 63)
    for mode in (0, 1):
        cfg = dict(n_channels=6, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=mode, tz_offset_s=0,
                   ref_time_s=1_528_000_000, fail_q=7.0)
        full = po.build_tables(st, b["channel"], b["mux"], b["time"], None, cfg)
        some = po.build_tables(st, b["channel"], b["mux"], t2, None, cfg)
        assert int(some["totals"][po.TOT_NO_TIME]) == int(gone.sum()) and (some["hour"][gone] == -1).all()
        keep = ~gone
        sub = {k: (v[keep] if isinstance(v, np.ndarray) else [x for x, g in zip(v, keep) if g]) for k, v in st.items()}
        only = po.build_tables(sub, b["channel"][keep], b["mux"][keep], b["time"][keep], None, cfg)
        assert np.array_equal(some["hour"][keep], only["hour"])
        assert np.array_equal(some["cmh_reads"], only["cmh_reads"]) and np.array_equal(some["hour_bases"], only["hour_bases"])
        assert np.array_equal(some["chan_reads"], full["chan_reads"])
        assert_tables_equal(c_oracle.tables(c_oracle.per_read(b["bases"], b["quals"], b["offsets"]), b["channel"],
                                            b["mux"], t2, None, cfg), some, "C oracle, mode %d" % mode)


def test_summary_reader_and_fastq_reader():
    hdr = "filename read_id run_id channel start_time duration num_events passes_filtering template_start " \
          "num_events_template template_duration num_called_template sequence_length_template mean_qscore_template " \
          "strand_score_template calibration_strand_genome_template a b c d e"
    def row(ch, start, length, q):
        f = ["f.fast5", "rid", "run", str(ch), str(start), "1.0", "10", "True", "0.0", "10", "1.0", "10", str(length),
             str(q), "0", "x", "1", "2", "3"]
        # the reader slices [3:-5]: element [3] of the slice must be the bases column, [-1] the mean-q column
        core = [str(ch), str(start), "1.0", str(length), "True", "0.0", "10", str(q)]
        return " ".join(["f", "r", "u"] + core + ["p1", "p2", "p3", "p4", "p5"])
    lines = [hdr, row(2, 3600.5 * 23, 100, 8.0), row(2, 3600.0 * 24.5, 200, 6.5), row(2, 3600.0 * 25.5, 300, 9.0),
             row(1, 10.0, 50, 7.0)]
    cd, ref_start = po.summary_reader(lines, n_channels=4)
    assert ref_start == 10.0
    assert cd["2"][0] == 3 and cd["2"][1] == 600 and cd["2"][2] == 1
    assert cd["1"][:3] == (1, 50, 0) and cd["3"] == ()
    fs = po.FastStats_data(cd, GCTable=[41.0, 0.0, 52.5])
    assert fs["PassSum"] == 4 and fs["GCVec"] == [41.0, 52.5]
    fq = b"@r1 a b ch=5\nACGT\n+\nIIII\n@r2 a b ch=6\nGGCC\n+\n!!!!\n"
    assert po.fastq_reader(fq) == [50.0, 100.0]
    with pytest.raises(ValueError):
        list(po.parse_fastq_records(b"@r\nACGT\n+\nII\n"))


def test_c_oracle_equals_python_oracle(c_oracle):
    for seed, mode in ((5, 0), (6, 1)):
        b = random_batch(seed, 1500, 120, n_channels=24, hours=50)
        a = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
        c = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=3)
        for k in ("length", "gc", "n", "qsum"):
            assert np.array_equal(a[k], c[k]), k
        assert all(int(x) == int(y) for x, y in zip(a["err"], c["err"]))
        np.testing.assert_allclose(a["mean_q"], c["mean_q"], rtol=1e-14, atol=0)
        cfg = dict(n_channels=24, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=mode, tz_offset_s=1800,
                   ref_time_s=1_528_000_000, fail_q=7.0)
        assert_tables_equal(c_oracle.tables(c, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg),
                            po.build_tables(a, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg))
    assert np.array_equal(c_oracle.err_lut(), po.err_lut())
    for lens, want in KAT["n50"]:
        assert c_oracle.n50(lens) == want


def test_oracles_against_committed_golden(c_oracle):
    g = np.load(os.path.join(HERE, "golden", "random_small.npz"))
    gt = json.load(open(os.path.join(HERE, "golden", "random_small_tables.json")))
    c = c_oracle.per_read(g["bases"], g["quals"], g["offsets"], nthreads=2)
    for k in ("length", "gc", "n", "qsum"):
        assert np.array_equal(c[k], g["exp_" + k]), k
    assert [int(e) for e in c["err"]] == [(int(h) << 32) + int(l) for h, l in zip(g["exp_err_hi"], g["exp_err_lo"])]
    for cfgk, tk in (("cfg0", "tables0"), ("cfg1", "tables1")):
        t = c_oracle.tables(c, g["channel"], g["mux"], g["time"], g["fail_flag"], gt[cfgk])
        for k in ("totals", "q_hist", "gc_hist", "hour_reads", "chan_fail", "cmh_bases", "len_hist_lin"):
            assert np.array_equal(np.asarray(t[k]), np.asarray(gt[tk][k], dtype=np.int64)), (tk, k)
        assert t["n50"] == gt[tk]["n50"]


def test_nano_layout_is_the_minion_32x16_map():
    rows = po.nano_layout()
    assert len(rows) == 32 and all(len(r) == 16 for r in rows)
    assert sorted(int(x) for r in rows for x in r) == list(range(1, 513))
    assert rows[0] == [str(x) for x in (1, 2, 3, 4, 5, 6, 7, 8, 40, 39, 38, 37, 36, 35, 34, 33)]        # seq_routines.py:589
    assert rows[4][:3] == ["449", "450", "451"] and rows[4][8] == "488"                                  # :593
    assert rows[31] == [str(x) for x in (89, 90, 91, 92, 93, 94, 95, 96, 128, 127, 126, 125, 124, 123, 122, 121)]  # :670


def test_constants_pinned_to_the_reference_source():
    """tests/golden/reference_constants.json holds the data constants the reference hard-codes on this path, extracted
    from its source text by tests/golden/make_reference_constants.py (the one anchor this build has in the reference
    itself, which ships no tests): the oracle, the host drop-in and the C ABI must agree with them."""
    import ctypes as C
    import hashlib
    import json
    import os
    from oracle import pypore_oracle as po
    from pypore_amd import _lib
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_constants.json")))
    lay = ref["nano_layout"]
    import importlib
    sr = importlib.import_module("pypore_amd.seq_routines")
    for layout in (po.nano_layout(), sr.nano_layout()):
        flat = [x for row in layout for x in row]
        assert (len(layout), len(layout[0]), len(flat)) == (lay["rows"], lay["cols"], lay["count"])
        assert hashlib.sha256(",".join(flat).encode()).hexdigest() == lay["sha256_of_comma_joined"]
        assert layout[0] == lay["first_row"] and layout[-1] == lay["last_row"]
    assert (sr.n_channels, sr.n_hours) == (ref["n_channels"], ref["n_hours"])
    q, g = ref["quality_hist"], ref["gc_hist"]
    assert _lib.N_QHIST == (q["end"] - q["start"]) // q["size"] and q["start"] == 6      # bins 6+j <= q < 7+j
    assert _lib.N_GCHIST == (g["end"] - g["start"]) // g["size"] and g["size"] == 5
    assert _lib.N_LEN_LIN == ref["length_hist"]["nbinsx"]
    assert ref["fail_below_q"] == ref["fail_below_q_summary"] == ref["fastq_kept_from_q"] == 7.0
    assert ref["fail_path_default_q"] == 6.99 and ref["muxes"] == [1, 2, 3, 4]
    cfg = _lib.Config()
    lib = _lib.load()
    assert lib.ppqc_config_default(C.byref(cfg)) == 0
    assert (cfg.n_channels, cfg.n_mux, cfg.n_time_bins, cfg.fail_q) == (ref["n_channels"], len(ref["muxes"]),
                                                                       ref["n_hours"], ref["fail_below_q"])
    assert po.default_config()["fail_q"] == ref["fail_below_q"]
