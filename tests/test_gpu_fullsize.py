"""BASELINE configs at FULL size on the GPU, checked through size-independent properties (the oracle cannot chew
10 Gbases inside a test): conservation against independent whole-buffer reductions done with torch on the same HBM
bytes, linearity under splitting the run into batches, N50 against numpy on the 10^6 lengths, and bit-exact oracle
parity on a random sample of reads (including the longest ones) whose bytes are regenerated on the host."""
import numpy as np
import pytest

from util import assert_tables_equal

pytestmark = pytest.mark.gpu


def _torch_counts(torch, tb, tq, nb):
    gc = nn = qs = 0
    step = 1 << 28
    for a in range(0, nb, step):
        b = tb[a:min(nb, a + step)]
        gc += int(((b == ord("G")) | (b == ord("C"))).sum())
        nn += int((b == ord("N")).sum())
        qs += int(tq[a:min(nb, a + step)].sum(dtype=torch.int64))
    return gc, nn, qs - 33 * nb


@pytest.mark.parametrize("name,n_reads,mu,n_channels,n_bins", [("config2", 1_000_000, 8.5172, 512, 48),
                                                               ("config4", 200_000, 10.1266, 512, 48),
                                                               ("config3 shard", 1_250_000, 7.8240, 3000, 72)])
def test_full_size_properties(c_oracle, name, n_reads, mu, n_channels, n_bins):
    import torch
    from pypore_amd import qc, synth
    seed = 20240601
    lens = synth.read_lengths(n_reads, mu, 1.1774, seed)
    off = synth.offsets_from_lengths(lens)
    nb, R = int(off[-1]), n_reads
    ch, mux, t = synth.metadata(R, n_channels, n_bins, seed)
    cfg = dict(n_channels=n_channels, n_mux=4, n_time_bins=n_bins, bin_seconds=3600, time_mode=0, tz_offset_s=0,
               ref_time_s=1_528_000_000, fail_q=7.0, max_reads_hint=R)
    with qc.QCContext(device=0, **cfg) as ctx:
        tb = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
        tq = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
        ctx.synth_fill_device(tb.data_ptr(), tq.data_ptr(), 0, nb, seed)
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        d = [torch.from_numpy(x).cuda() for x in (ch.view(np.int32), mux, t)]
        ctx.submit_device(R, nb, tb.data_ptr(), tq.data_ptr(), d_off.data_ptr(), *[x.data_ptr() for x in d])
        s1 = ctx.read_stats()
        t1 = ctx.finalize()
        # linearity: the same run as three device batches cut at read boundaries (16-byte aligned cuts)
        ctx.reset()
        cuts = [0]
        for frac in (0.31, 0.77):
            r = int(R * frac)
            while int(off[r]) % 16:
                r += 1
            cuts.append(r)
        cuts.append(R)
        subs = []
        for a, e in zip(cuts[:-1], cuts[1:]):
            so = torch.from_numpy((off[a:e + 1] - off[a]).view(np.int64)).cuda()
            subs.append(so)
            b0 = int(off[a])
            ctx.submit_device(e - a, int(off[e]) - b0, tb.data_ptr() + b0, tq.data_ptr() + b0, so.data_ptr(),
                              d[0].data_ptr() + 4 * a, d[1].data_ptr() + a, d[2].data_ptr() + 8 * a)
        s2 = ctx.read_stats()
        t2 = ctx.finalize()
        gc, nn, qs = _torch_counts(torch, tb, tq, nb)
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), (name, k)
    assert_tables_equal(t1, t2, name)
    # conservation
    assert np.array_equal(s1["length"], lens)
    assert int(s1["gc"].sum(dtype=np.int64)) == gc and int(s1["n"].sum(dtype=np.int64)) == nn
    assert int(s1["qsum"].sum()) == qs
    tot = t1["totals"]
    assert (tot[0], tot[1], tot[3], tot[4], tot[5]) == (R, nb, gc, nn, qs)
    assert int(tot[2]) == int((s1["flags"] & 1).sum())
    assert int(t1["chan_reads"].sum()) == R and int(t1["chan_bases"].sum()) == nb
    assert int(t1["hour_reads"].sum()) == int(t1["cmh_reads"].sum()) == int(tot[6]) == int((mux >= 1).sum())
    assert int(t1["len_hist_lin"].sum()) == R == int(t1["len_hist_log2"].sum())
    assert np.array_equal(t1["chan_reads"], np.bincount(ch - 1, minlength=n_channels))
    assert np.array_equal(t1["chan_bases"], np.bincount(ch - 1, weights=lens.astype(np.float64), minlength=n_channels).astype(np.int64))
    assert t1["n50"] == c_oracle.n50(lens) and t1["max_len"] == int(lens.max())
    # oracle parity on a sample of reads, bytes regenerated from (seed, index): the longest reads and random ones
    rng = np.random.default_rng(1)
    pick = np.unique(np.concatenate([np.argsort(lens)[-4:], rng.integers(0, R, 150), [0, R - 1]]))
    hb, hq, ho = [], [], [0]
    for i in pick:
        b, q = synth.stream_bytes(int(off[i]), int(lens[i]), seed)
        hb.append(b); hq.append(q); ho.append(ho[-1] + int(lens[i]))
    want = c_oracle.per_read(np.concatenate(hb), np.concatenate(hq), np.asarray(ho, np.uint64), nthreads=4)
    for k in ("length", "gc", "n", "qsum"):
        assert np.array_equal(s1[k][pick], want[k]), (name, k)
    assert [(int(h) << 32) + int(l) for h, l in zip(s1["err_hi"][pick], s1["err_lo"][pick])] == [int(e) for e in want["err"]]
    np.testing.assert_allclose(s1["mean_q"][pick], want["mean_q"], rtol=1e-12)


def test_config3_whole_job_on_one_gpu(c_oracle):
    """BASELINE config 3 WHOLE on one GPU: 10^7 reads, 3000 channels x 4 muxes x 72 hourly bins, about 50 Gbases (100 GB of
    the 288 GB resident).  Conservation against torch reductions over the same HBM bytes, N50 against the oracle over the
    10^7 lengths, bit-exact oracle parity on sampled reads, and 'eight contiguous shards merged on the host == the whole
    job' for every mergeable table (what the 8-GPU run of the driver computes with RCCL instead)."""
    import torch
    from pypore_amd import qc, synth
    from pypore_amd.merge import merge_tables_host
    c3 = synth.CONFIGS["config3"]
    seed, R = 20240601, c3["n_reads"]
    lens = synth.read_lengths(R, c3["mu"], c3["sigma"], seed)
    off = synth.offsets_from_lengths(lens)
    nb = int(off[-1])
    ch, mux, t = synth.metadata(R, c3["n_channels"], c3["hours"], seed)
    cfg = dict(n_channels=c3["n_channels"], n_mux=4, n_time_bins=c3["n_time_bins"], bin_seconds=3600, time_mode=0,
               tz_offset_s=0, ref_time_s=1_528_000_000, fail_q=7.0, max_reads_hint=R)
    with qc.QCContext(device=0, **cfg) as ctx:
        tb = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
        tq = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
        ctx.synth_fill_device(tb.data_ptr(), tq.data_ptr(), 0, nb, seed)
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        d = [torch.from_numpy(x).cuda() for x in (ch.view(np.int32), mux, t)]
        ctx.submit_device(R, nb, tb.data_ptr(), tq.data_ptr(), d_off.data_ptr(), *[x.data_ptr() for x in d])
        s1 = ctx.read_stats()
        whole = ctx.finalize()
        # eight contiguous shards, one after the other on the same context (cuts moved to 16-byte aligned read starts)
        cuts = [0]
        for k in range(1, 8):
            r = k * R // 8
            while int(off[r]) % 16:
                r += 1
            cuts.append(r)
        cuts.append(R)
        shards, keep = [], []
        for a, e in zip(cuts[:-1], cuts[1:]):
            ctx.reset()
            so = torch.from_numpy((off[a:e + 1] - off[a]).view(np.int64)).cuda()
            keep.append(so)
            b0 = int(off[a])
            ctx.submit_device(e - a, int(off[e]) - b0, tb.data_ptr() + b0, tq.data_ptr() + b0, so.data_ptr(),
                              d[0].data_ptr() + 4 * a, d[1].data_ptr() + a, d[2].data_ptr() + 8 * a)
            shards.append(ctx.finalize())
        gc, nn, qs = _torch_counts(torch, tb, tq, nb)
    merged = merge_tables_host(shards)
    for k in merged:
        if k == "len_hist_lin":
            continue
        assert np.array_equal(np.asarray(merged[k]).astype(np.int64), np.asarray(whole[k]).astype(np.int64)), k
    assert merged["max_len"] == whole["max_len"] and merged["min_pos_len"] == whole["min_pos_len"]
    # conservation
    assert np.array_equal(s1["length"], lens)
    tot = whole["totals"]
    assert (int(tot[0]), int(tot[1]), int(tot[3]), int(tot[4]), int(tot[5])) == (R, nb, gc, nn, qs)
    assert int(s1["gc"].sum(dtype=np.int64)) == gc and int(s1["qsum"].sum()) == qs
    assert int(tot[2]) == int((s1["flags"] & 1).sum())
    assert int(whole["chan_reads"].sum()) == R and int(whole["chan_bases"].sum()) == nb
    assert int(whole["hour_reads"].sum()) == int(whole["cmh_reads"].sum()) == int(tot[6]) == int((mux >= 1).sum())
    assert int(whole["cmh_bases"].sum()) == int(whole["hour_bases"].sum()) == int(lens[mux >= 1].sum(dtype=np.int64))
    assert np.array_equal(whole["chan_reads"], np.bincount(ch - 1, minlength=c3["n_channels"]))
    cell = ((ch.astype(np.int64) - 1) * 4 + (mux.astype(np.int64) - 1)) * c3["n_time_bins"] + (t - 1_528_000_000) // 3600
    want_cells = np.bincount(cell[mux >= 1], minlength=c3["n_channels"] * 4 * c3["n_time_bins"])
    assert np.array_equal(np.asarray(whole["cmh_reads"]).reshape(-1), want_cells)
    assert whole["n50"] == c_oracle.n50(lens) and whole["max_len"] == int(lens.max())
    print("config 3 whole: %d reads, %d bases, N50 %d, %d fail" % (R, nb, whole["n50"], int(tot[2])))
    rng = np.random.default_rng(3)
    pick = np.unique(np.concatenate([np.argsort(lens)[-3:], rng.integers(0, R, 120), [0, R - 1]]))
    hb, hq, ho = [], [], [0]
    for i in pick:
        b, q = synth.stream_bytes(int(off[i]), int(lens[i]), seed)
        hb.append(b); hq.append(q); ho.append(ho[-1] + int(lens[i]))
    want = c_oracle.per_read(np.concatenate(hb), np.concatenate(hq), np.asarray(ho, np.uint64), nthreads=4)
    for k in ("length", "gc", "n", "qsum"):
        assert np.array_equal(s1[k][pick], want[k]), k
    assert [(int(h) << 32) + int(l) for h, l in zip(s1["err_hi"][pick], s1["err_lo"][pick])] == [int(e) for e in want["err"]]
