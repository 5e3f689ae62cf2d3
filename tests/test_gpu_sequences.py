"""Randomised CALL SEQUENCES through the C ABI: random interleavings of every way reads enter a context (host CSR
batches on either staging slot, device batches, sequencing_summary rows, FASTQ text from the host and from the device),
of every way results leave it (finalize with copies or views, the phased finalize with the sort-based and the select
N50, the time_mode 1 phases of the multi-GPU path, partial read_stats, read_hours) and of what changes state in between
(set_ref_time, sync, reset, calls out of order that must fail and leave the context usable).  After every observation
each table is compared with the oracle over the concatenation of everything submitted since the last reset.
The kernels are fuzzed elsewhere (tools/fuzz_parity.py); this fuzzes the context's bookkeeping: which stream holds what,
which results are stale, what a batch left at submit time."""
import numpy as np
import pytest

from util import random_batch, assert_tables_equal

pytestmark = pytest.mark.gpu

N_SEQUENCES = 400
I64_MIN = np.iinfo(np.int64).min


class Model:
    """Everything submitted since the last reset, as the oracle wants it."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.cols = dict(length=[], gc=[], n=[], qsum=[], err=[], mean_q=[], is_record=[], channel=[], mux=[], time=[], ff=[])

    def add(self, stats, channel, mux, t, ff=None, record=False):
        n = len(channel)
        c = self.cols
        c["length"].append(np.asarray(stats["length"], np.uint32)); c["gc"].append(np.asarray(stats["gc"], np.uint32))
        c["n"].append(np.asarray(stats["n"], np.uint32)); c["qsum"].append(np.asarray(stats["qsum"], np.int64))
        c["err"].append([int(e) for e in stats["err"]]); c["mean_q"].append(np.asarray(stats["mean_q"], np.float64))
        c["is_record"].append(np.full(n, 1 if record else 0, np.uint8))
        c["channel"].append(np.asarray(channel, np.uint32)); c["mux"].append(np.asarray(mux, np.uint8))
        c["time"].append(np.asarray(t, np.int64)); c["ff"].append(np.zeros(n, np.uint8) if ff is None else np.asarray(ff, np.uint8))

    @property
    def n(self):
        return sum(len(x) for x in self.cols["channel"])

    def cat(self, k):
        v = self.cols[k]
        if k == "err":
            return [e for part in v for e in part]
        return np.concatenate(v) if v else np.zeros(0, dict(mean_q=np.float64, qsum=np.int64, time=np.int64).get(k, np.uint32))

    def tables(self, co, cfg):
        stats = {k: self.cat(k) for k in ("length", "gc", "n", "qsum", "err", "mean_q", "is_record")}
        return co.tables(stats, self.cat("channel"), self.cat("mux").astype(np.uint8), self.cat("time"),
                         self.cat("ff").astype(np.uint8), cfg)


def _fastq_text(rng, n, n_channels, ref, hours):
    alph = np.frombuffer(b"ACGTNacgtSs", np.uint8)
    out = []
    import time as _t
    for i in range(n):
        L = int(rng.choice([0, 1, 16, 17, 300, int(rng.integers(1, 2500))]))
        s, q = bytes(rng.choice(alph, L)), bytes(rng.integers(33, 100, L, dtype=np.uint8))
        stamp = _t.strftime("%Y-%m-%dT%H:%M:%SZ", _t.gmtime(ref + int(rng.integers(0, hours * 3600)))).encode()
        out.append(b"@r%d runid=z read=%d ch=%d start_time=%s\n%s\n+\n%s\n" % (i, i, int(rng.integers(1, n_channels + 1)), stamp, s, q))
    return b"".join(out)


def _records(rng, n, n_channels, ref, hours):
    length = rng.integers(0, 30000, n).astype(np.uint32)
    mean_q = np.round(rng.normal(9.0, 2.5, n), 3)
    gc = (length * rng.uniform(0.2, 0.7, n)).astype(np.uint32)
    channel = rng.integers(1, n_channels + 1, n).astype(np.uint32)
    mux = rng.integers(1, 5, n).astype(np.uint8)
    t = (ref + rng.integers(0, hours * 3600, n)).astype(np.int64)
    return length, mean_q, gc, channel, mux, t


def _one_sequence(seed, qc, co):
    import torch
    from pypore_amd import fastq_io
    from pypore_amd._lib import PPQCError
    rng = np.random.default_rng(10_000 + seed)
    ref = 1_528_000_000
    nch = int(rng.choice([8, 64, 512]))
    cfg = dict(n_channels=nch, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=int(rng.integers(0, 2)),
               tz_offset_s=int(rng.choice([0, 3600, -18000])), ref_time_s=ref, fail_q=7.0)
    compat = int(rng.integers(0, 2))
    hours = 40
    model = Model()
    log = []
    dirty = True        # nothing finalised since the context was created, reset or given reads (library state 0)
    keep_alive = []     # device tensors of device batches: alive until the context has been synchronised

    def expect(tables, what):
        want = model.tables(co, dict(cfg, compat_flags=compat))
        try:
            assert_tables_equal(tables, want, what)
        except AssertionError as e:
            raise AssertionError("sequence %d (%s), calls %s: %s" % (seed, cfg, log, e))

    with qc.QCContext(device=0, compat_flags=compat, **cfg) as ctx:
        for step in range(int(rng.integers(5, 13))):
            op = rng.choice(["csr", "csr", "device", "records", "fastq", "fastq_dev", "finalize", "finalize", "view", "phased",
                             "select", "mode1", "read_stats", "ref", "sync", "reset", "bad_order", "empty"])
            log.append(str(op))
            if op in ("csr", "device", "records", "fastq", "fastq_dev", "reset"):
                dirty = True
            if op in ("finalize", "view", "phased", "select", "mode1"):
                dirty = False
            if op in ("csr", "device"):
                b = random_batch(int(rng.integers(1 << 30)), int(rng.integers(1, 500)), int(rng.choice([30, 300, 3000])),
                                 n_channels=nch, hours=hours, edge=bool(rng.integers(0, 2)))
                if rng.random() < 0.2:
                    b["time"][rng.random(len(b["time"])) < 0.1] = I64_MIN
                ff = b["fail_flag"] if rng.random() < 0.7 else None
                st = co.per_read(b["bases"], b["quals"], b["offsets"], nthreads=2)
                if ff is not None:      # flagged reads keep the reference's defaults (len 0, q 6.99, gc 0)
                    fl = ff.astype(bool)
                    for k in ("length", "gc", "n", "qsum"):
                        st[k] = np.where(fl, 0, st[k])
                    st["err"] = [0 if f else int(e) for f, e in zip(fl, st["err"])]
                    st["mean_q"] = np.where(fl, 6.99, st["mean_q"])
                if op == "csr":
                    ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], ff, slot=int(rng.integers(0, 2)))
                else:
                    nb = int(b["offsets"][-1])
                    pad = lambda a: torch.from_numpy(np.concatenate([a, np.zeros(64, a.dtype)])).cuda()
                    d = [pad(b["bases"]), pad(b["quals"]), torch.from_numpy(b["offsets"].view(np.int64)).cuda(),
                         torch.from_numpy(b["channel"].view(np.int32)).cuda(), torch.from_numpy(b["mux"]).cuda(),
                         torch.from_numpy(b["time"]).cuda()]
                    if ff is not None:
                        d.append(torch.from_numpy(ff).cuda())
                    torch.cuda.synchronize()
                    ctx.submit_device(len(b["channel"]), nb, *[x.data_ptr() for x in d])
                    keep_alive.append(d)
                model.add(st, b["channel"], b["mux"], b["time"], ff)
            elif op == "records":
                length, mean_q, gc, channel, mux, t = _records(rng, int(rng.integers(1, 400)), nch, ref, hours)
                ctx.submit_records(length, mean_q, channel, mux, t, gc)
                n = len(length)
                model.add(dict(length=length, gc=gc, n=np.zeros(n), qsum=np.zeros(n), err=[0] * n, mean_q=mean_q), channel, mux, t,
                          record=True)
            elif op in ("fastq", "fastq_dev"):
                text = _fastq_text(rng, int(rng.integers(1, 200)), nch, ref, hours)
                p = fastq_io.pack_fastq(text)
                mux = int(rng.integers(1, 5))
                if op == "fastq":
                    assert ctx.submit_fastq(text, slot=int(rng.integers(0, 2)), mux=mux) == p["n_reads"]
                else:
                    dt = torch.from_numpy(np.concatenate([np.frombuffer(text, np.uint8), np.zeros(64, np.uint8)])).cuda()
                    torch.cuda.synchronize()
                    assert ctx.submit_fastq_device(dt.data_ptr(), len(text), mux=mux) == p["n_reads"]
                    keep_alive.append(dt)
                st = co.per_read(p["bases"], p["quals"], p["offsets"], nthreads=2)
                model.add(st, p["channel"], np.full(p["n_reads"], mux, np.uint8), p["start_time"])
            elif op == "empty":
                ctx.submit(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint32),
                           np.zeros(0, np.uint8), np.zeros(0, np.int64))
            elif op == "finalize":
                expect(ctx.finalize(), "finalize")
            elif op == "view":
                v = ctx.finalize(copy=False)
                expect({k: (np.array(x) if isinstance(x, np.ndarray) else x) for k, x in v.items()}, "finalize view")
            elif op in ("phased", "select", "mode1"):
                ctx.finalize_extremes()
                if rng.random() < 0.5:
                    tmin, tmax = ctx.time_range()
                    t_all, ch_all = model.cat("time"), model.cat("channel")
                    ok = (ch_all >= 1) & (ch_all <= nch) & (t_all != I64_MIN)
                    assert (tmin, tmax) == ((int(t_all[ok].min()), int(t_all[ok].max())) if ok.any() else
                                            (np.iinfo(np.int64).max, I64_MIN)), (seed, log)
                if cfg["time_mode"] == 1 and rng.random() < 0.4:     # takes effect in this finalize's tables (mode 1)
                    cfg["ref_time_s"] = ref + int(rng.integers(-3, 4)) * 1800
                    ctx.set_ref_time(cfg["ref_time_s"])
                if op == "mode1" and cfg["time_mode"] == 1:
                    for ph in range(3):
                        ctx.mode1_dist_phase(ph)
                ctx.finalize_tables()
                if op == "select":
                    ctx.n50_select()
                else:
                    ctx.finalize_n50()
                expect(ctx.fetch_tables(), op)
                if model.n and rng.random() < 0.5:
                    assert len(ctx.read_hours()) == model.n
            elif op == "read_stats":
                n = model.n
                if n:
                    a = int(rng.integers(0, n))
                    e = int(rng.integers(a, n + 1))
                    got = ctx.read_stats(a, e - a)
                    rec = model.cat("is_record")[a:e].astype(bool)
                    for k, mk in (("length", "length"), ("gc", "gc"), ("n", "n"), ("qsum", "qsum")):
                        assert np.array_equal(got[k], model.cat(mk)[a:e].astype(got[k].dtype)), (seed, log, k)
                    gerr = [(int(h) << 32) + int(l) for h, l in zip(got["err_hi"], got["err_lo"])]
                    assert gerr == model.cat("err")[a:e], (seed, log)
                    assert np.array_equal(got["mean_q"][rec], model.cat("mean_q")[a:e][rec])
            elif op == "ref":
                cfg["ref_time_s"] = ref + int(rng.integers(-3, 4)) * 1800
                ctx.set_ref_time(cfg["ref_time_s"])
            elif op == "sync":
                ctx.sync()
                keep_alive.clear()
            elif op == "reset":
                ctx.reset()
                model.clear()
                keep_alive.clear()
            elif op == "bad_order":
                if dirty:     # nothing finalised since the last change: these must fail and change nothing
                    with pytest.raises(PPQCError):
                        ctx.fetch_tables()
                    with pytest.raises(PPQCError):
                        ctx.finalize_tables()
        expect(ctx.finalize(), "last finalize")
        ctx.sync()
    return len(log)


def test_random_call_sequences(qc_mod, c_oracle):
    calls = 0
    for seed in range(N_SEQUENCES):
        calls += _one_sequence(seed, qc_mod, c_oracle)
    assert calls > 5 * N_SEQUENCES
