"""The RCCL merge path of bench.py --gpus N, exercised on the one GPU a test box has: a 1-rank "nccl" group runs
finalize_distributed directly on the library's device buffers: the early merge (extremes all-reduce and the three
select rounds on a side stream beside the per-read kernel, one table SUM behind it), the phased merge (four collectives
behind the kernel) and the gather variant (all-gather of the lengths, sort on every rank).  The N-rank arithmetic is covered by the 2-rank
gloo test (tests/test_dist_gloo.py)."""
import socket

import numpy as np
import pytest

from util import random_batch, assert_tables_equal

pytestmark = pytest.mark.gpu


def test_finalize_distributed_one_rank_nccl_and_zero_copy_views(c_oracle):
    import torch
    import torch.distributed as dist
    from pypore_amd import qc
    from pypore_amd.merge import finalize_distributed
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        cfg = dict(n_channels=3000, n_mux=4, n_time_bins=72, bin_seconds=3600, time_mode=0, tz_offset_s=0,
                   ref_time_s=1_528_000_000, fail_q=7.0)
        b = random_batch(5, 20000, 300, n_channels=3000, hours=75)
        with qc.QCContext(device=0, **cfg) as ctx:
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
            merged = finalize_distributed(ctx, dist, torch)                  # the merges beside the per-read kernel (default)
            assert ctx.external_stream == torch.cuda.current_stream().cuda_stream
            assert ctx.index_early() is False                                # consumed by the finalize
            ctx.reset()
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
            assert ctx.index_early() is True
            phased = finalize_distributed(ctx, dist, torch, early=False)     # four collectives behind the kernel
            assert_tables_equal(merged, phased, "early vs phased merge")
            ctx.reset()                                                      # an empty shard takes the early path too
            empty = finalize_distributed(ctx, dist, torch)
            assert empty["totals"].sum() == 0 and empty["n50"] == 0 and empty["max_len"] == 0
            ctx.reset()
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
            gathered = finalize_distributed(ctx, dist, torch, n50="gather")   # all-gather + sort on every rank
            assert_tables_equal(merged, gathered, "select vs gather")
            views, d = ctx.device_tables()
            assert d.sel_buf == d.sum_buf + 8 * d.n_sum and d.n_sel == 2048      # [sum | sel] is one all-reduce
            ctx.use_own_stream()
            t_sum = torch.as_tensor(views["sum"], device="cuda")
            assert t_sum.data_ptr() == d.sum_buf and t_sum.numel() == d.n_sum      # in-place collectives need zero copy
            lens = torch.as_tensor(views["lengths"], device="cuda")
            assert lens.data_ptr() == d.lengths and lens.numel() == len(b["channel"])
            ctx.reset()
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
            single = ctx.finalize()
        assert_tables_equal(merged, single, "1-rank merge")
        want = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=4)
        assert_tables_equal(merged, c_oracle.tables(want, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg))
    finally:
        dist.destroy_process_group()


def test_mode1_phases_equal_the_sort_based_hours(qc_mod, c_oracle):
    """time_mode 1 from the three per-channel tables (ppqc_mode1_dist_phase, no collectives: one rank) against the
    sort-based path of ppqc_finalize_tables and the oracle's sequential rule: per-read hour bins and every table."""
    from pypore_amd._lib import PPQCError
    qc = qc_mod
    for seed, n, tz, nch in ((21, 20000, 0, 512), (22, 3000, -18000, 24), (23, 40, 7200, 6), (24, 50000, 0, 3000)):
        cfg = dict(n_channels=nch, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=tz,
                   ref_time_s=1_528_000_000, fail_q=7.0)
        b = random_batch(seed, n, 120, n_channels=nch, hours=60)
        t = b["time"].copy()
        t[::17] = t[0]                                    # ties, some at a channel's earliest start
        t[5::29] = np.iinfo(np.int64).min                 # reads without a start time
        args = (b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], t, b["fail_flag"])
        with qc.QCContext(device=0, **cfg) as ctx:
            ctx.submit(*args)
            sort_tables = ctx.finalize()
            sort_hours = ctx.read_hours()
            ctx.reset()
            ctx.submit(*args)
            ctx.finalize_extremes()
            with pytest.raises(PPQCError, match="order 0, 1, 2"):
                ctx.mode1_dist_phase(1)
            for ph in range(3):
                ctx.mode1_dist_phase(ph)
            ctx.finalize_tables()
            ctx.finalize_n50()
            got = ctx.fetch_tables()
            assert np.array_equal(ctx.read_hours(), sort_hours), seed
            assert_tables_equal(got, sort_tables, "mode-1 tables vs sort, seed %d" % seed)
            # the host statement of the same phases (what the gloo tests run across two ranks)
            from pypore_amd import merge
            ref_hours, oob = merge.mode1_hours_reference(b["channel"], t, cfg, lambda a: None, lambda a: None)
            assert oob == 0 and np.array_equal(ref_hours, sort_hours)
        want = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=4)
        assert_tables_equal(got, c_oracle.tables(want, b["channel"], b["mux"], t, b["fail_flag"], cfg))
    # a read 70 days behind the reference time: outside the window of the per-day tables, and the run says so
    cfg = dict(n_channels=4, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0, ref_time_s=10**9)
    with qc.QCContext(device=0, **cfg) as ctx:
        ctx.submit(np.frombuffer(b"ACGTACGT", np.uint8), np.frombuffer(b"IIIIIIII", np.uint8), np.array([0, 4, 8], np.uint64),
                   np.array([1, 1], np.uint32), np.array([1, 1], np.uint8), np.array([10**9, 10**9 + 86400 * 70], np.int64))
        ctx.finalize_extremes()
        for ph in range(3):
            ctx.mode1_dist_phase(ph)
        ctx.finalize_tables()
        ctx.finalize_n50()
        with pytest.raises(PPQCError, match="day window"):
            ctx.fetch_tables()


def test_group_finalize_without_torch_collectives(qc_mod, c_oracle):
    """ppqc_group_*: the library's own RCCL merge for one process that drives several GPUs (librccl through dlopen, no torch
    in the call path).  One GPU here, so a group of one context -- every collective runs, over a single rank -- in both
    time modes, against ppqc_finalize on the same reads and the oracle; a second finalize of the group after more reads."""
    from pypore_amd._lib import PPQCError
    for mode in (0, 1):
        cfg = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=mode, tz_offset_s=3600,
                   ref_time_s=1_528_000_000, fail_q=7.0)
        b = random_batch(41 + mode, 8000, 500, hours=46)
        args = (b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["time"], b["fail_flag"])
        with qc_mod.QCContext(device=0, **cfg) as ctx, qc_mod.QCContext(device=0, **cfg) as single:
            single.submit(*args)
            want = single.finalize()
            with qc_mod.Group([ctx]) as grp:
                ctx.submit(*args)
                got = grp.finalize()
                assert_tables_equal(got, want, "group of one, time_mode %d" % mode)
                ctx.submit(*args)                       # more reads behind a group finalize
                single.submit(*args)
                assert_tables_equal(grp.finalize(), single.finalize(), "second group finalize")
            with pytest.raises(PPQCError, match="one GPU"):
                qc_mod.Group([ctx, single])
        w = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=4)
        assert_tables_equal(got, c_oracle.tables(w, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg))


def test_two_ranks_on_one_gpu_over_gloo():
    """tools/dist_rehearsal.py: world_size 2 (two processes sharing the one GPU, gloo because RCCL refuses two ranks on a
    device): early, phased and gather merge and an empty shard on one rank, every rank's tables against the host merge
    of the single-shard tables and the oracle's N50 over all lengths."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "tools", "dist_rehearsal.py")],
                       capture_output=True, text=True, timeout=170, cwd=root)
    if r.returncode != 0:   # the ranks' own words (a hang ends in the watchdog's traceback), kept where a GPU run leaves files
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        open(os.path.join(root, "gpurun_out", "dist_rehearsal_fail.log"), "w").write(r.stdout + "\n---- stderr\n" + r.stderr)
    assert r.returncode == 0 and "dist rehearsal ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _device_job(torch, qc, synth, ctx, n_reads, mu, n_channels, hours, seed, tweak_times=False):
    """A synthetic job resident in HBM: (lens, off, device arrays) -- bytes from the HBM generator, metadata from numpy."""
    lens = synth.read_lengths(n_reads, mu, 1.1774, seed)
    off = synth.offsets_from_lengths(lens)
    nb = int(off[-1])
    ch, mux, t = synth.metadata(n_reads, n_channels, hours, seed)
    if tweak_times:
        t = t.copy()
        t[::17] = t[0]                                    # ties, some at a channel's earliest start
        t[5::29] = np.iinfo(np.int64).min                 # reads without a start time
    tb = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
    tq = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
    ctx.synth_fill_device(tb.data_ptr(), tq.data_ptr(), 0, nb, seed)
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    d_ch, d_mux, d_t = (torch.from_numpy(x).cuda() for x in (ch.view(np.int32), mux, t))
    ctx.sync()
    torch.cuda.synchronize()
    return dict(lens=lens, off=off, nb=nb, R=n_reads, tb=tb, tq=tq, d_off=d_off, d_ch=d_ch, d_mux=d_mux, d_t=d_t,
                ch=ch, mux=mux, t=t)


def _submit_shard(torch, ctx, job, a, e, keep):
    """Reads [a, e) of a device-resident job as one device batch (off[a] must be a multiple of 16)."""
    off = job["off"]
    if e == a:
        return                                            # an empty shard submits nothing
    b0 = int(off[a])
    assert b0 % 16 == 0
    so = torch.from_numpy((off[a:e + 1] - off[a]).view(np.int64)).cuda()
    keep.append(so)
    if True:
        ctx.submit_device(e - a, int(off[e]) - b0, job["tb"].data_ptr() + b0, job["tq"].data_ptr() + b0, so.data_ptr(),
                          job["d_ch"].data_ptr() + 4 * a, job["d_mux"].data_ptr() + a, job["d_t"].data_ptr() + 8 * a)


FLOWS = (("early", 0, {}), ("phased", 0, dict(early=False)), ("gather", 0, dict(n50="gather")), ("time_mode 1", 1, {}))
SLEEP_CYCLES = 40_000_000     # torch.cuda._sleep: the stream is held while the host enqueues the whole chain behind it


def _nccl_group_of_one(torch, dist):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))


def _snapshot_flow(torch, dist, label, mode, kw):
    """One merge flow on torch's current stream: the snapshots of a run whose chain is enqueued behind a sleep kernel
    against those of a run that synchronises the device in front of every collective.  Returns (tables equal, number of
    collectives, indices of the collectives whose snapshots differ, issuing streams)."""
    from pypore_amd import qc, synth
    from pypore_amd.merge import finalize_distributed
    from stream_collectives import SpyDist
    cfg = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=mode, tz_offset_s=0,
               ref_time_s=1_528_000_000, fail_q=7.0, max_reads_hint=400_000)
    with qc.QCContext(device=0, **cfg) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        job = _device_job(torch, qc, synth, ctx, 400_000, 7.3, 512, 47, 77, tweak_times=mode == 1)
        other = _device_job(torch, qc, synth, ctx, 150_000, 6.0, 512, 30, 78)
        assert 2 * job["nb"] > 1_000_000_000

        def run(j, spy, sleep):
            ctx.reset()
            if sleep:
                torch.cuda._sleep(SLEEP_CYCLES)
            ctx.submit_device(j["R"], j["nb"], j["tb"].data_ptr(), j["tq"].data_ptr(), j["d_off"].data_ptr(),
                              j["d_ch"].data_ptr(), j["d_mux"].data_ptr(), j["d_t"].data_ptr())
            return finalize_distributed(ctx, spy, torch, **kw)

        ref = SpyDist(dist, torch, drain=True)
        want = run(job, ref, False)
        run(other, SpyDist(dist, torch, drain=True), False)     # leaves another job's values in every buffer
        spy = SpyDist(dist, torch)
        got = run(job, spy, True)
        try:
            assert_tables_equal(got, want, label)
            same = True
        except AssertionError:
            same = False
        assert len(spy.snaps) == len(ref.snaps) and [a for a, _ in spy.snaps] == [b for b, _ in ref.snaps]
        bad = [k for k, ((_, a), (_, b)) in enumerate(zip(spy.snaps, ref.snaps)) if not torch.equal(a, b)]
        return same, len(spy.snaps), bad, {a for a, _ in spy.snaps}


def test_every_collective_finds_its_producers_in_stream_order():
    """The stream-order contract of DESIGN section 5, checked without any host drain: in front of every collective of
    finalize_distributed a device-side snapshot of its tensor is enqueued on the stream the collective is issued on
    (SpyDist), while the context's stream is held by a sleep kernel in front of a 1.2 GB batch -- so the whole chain
    (per-read kernels, index kernels on the auxiliary stream, phases, collectives, snapshots) is enqueued before any of
    it runs, and a producer that is not ordered in front of its collective leaves a snapshot of the previous run's
    (different) job.  Every snapshot must equal the one of a run that synchronises the device in front of each
    collective.  Early, phased, gather and time_mode 1 merges, on torch's default stream and on a stream of its own;
    and the check itself is checked: with ppqc_stream_wait_index taken out, the early flow must be caught."""
    import torch
    import torch.distributed as dist
    from pypore_amd import qc
    _nccl_group_of_one(torch, dist)
    try:
        for own_stream in (False, True):
            with torch.cuda.stream(torch.cuda.Stream() if own_stream else torch.cuda.current_stream()):
                for label, mode, kw in FLOWS:
                    same, n, bad, streams = _snapshot_flow(torch, dist, label, mode, kw)
                    assert same and n >= 2 and not bad, \
                        "%s (own stream %s): collectives %s of %d found tensors their producers had not finished" % (
                            label, own_stream, bad, n)
                    # early: side-stream collectives + the table SUM; phased: the cells' all-reduce on its own stream
                    assert len(streams) == (2 if label in ("early", "phased") else 1)
        keep = qc.QCContext.stream_wait_index
        qc.QCContext.stream_wait_index = lambda self, handle: None          # negative control: a wait that is missing
        # (HIP maps its streams onto a few hardware queues -- GPU_MAX_HW_QUEUES, four here -- and two streams that share
        # one run in submission order whatever their events say: a side stream that happens to share the queue of the held
        # stream, or of the context's auxiliary stream with the index kernel waiting on it, finds its producers finished by
        # accident, and no check from outside can tell.  Which streams share depends on everything the process has created
        # so far (seen: blind after a time_mode 1 context had passed its streams on), so the control is run with a fresh
        # side stream (merge._beside_stream probes again, among fresh streams) and a fresh stream to hold until one pair
        # does not share.)
        from pypore_amd import merge
        tried = []
        side_kept = dict(merge._BESIDE)
        try:
            for _ in range(6):
                merge._BESIDE.clear()
                with torch.cuda.stream(torch.cuda.Stream()):
                    same, n, bad, _ = _snapshot_flow(torch, dist, *FLOWS[0])
                tried.append(bad)
                if bad and 0 in bad:
                    break
        finally:
            qc.QCContext.stream_wait_index = keep
            merge._BESIDE.clear()
            merge._BESIDE.update(side_kept)
        assert tried[-1] and 0 in tried[-1], "the snapshot check did not notice a missing wait (%s of %d)" % (tried, n)
    finally:
        dist.destroy_process_group()


def test_the_side_stream_runs_beside_the_current_one():
    """merge._beside_stream: the stream the merges beside the per-read kernel are issued on must not share the current
    stream's hardware queue (HIP maps streams onto a few; two streams on one queue run in submission order).  Holds the
    current stream with a sleep kernel and checks that an event on the chosen stream completes long before the sleep
    does -- on torch's default stream and on fresh ones, after the process has created a few dozen other streams."""
    import torch
    from pypore_amd import merge
    torch.cuda.set_device(0)
    junk = [torch.cuda.Stream() for _ in range(24)]     # whatever else a process may have created
    kept = dict(merge._BESIDE)
    try:
        for own in (False, True, True):
            with torch.cuda.stream(torch.cuda.Stream() if own else torch.cuda.current_stream()):
                merge._BESIDE.clear()
                cur = torch.cuda.current_stream()
                side = merge._beside_stream(torch)
                assert side.cuda_stream != cur.cuda_stream and merge._beside_stream(torch) is side   # probed once
                t0, t_cur, t_side = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                torch.cuda.synchronize()
                t0.record(cur)
                torch.cuda._sleep(2_000_000)
                t_cur.record(cur)
                side.wait_event(t0)
                t_side.record(side)
                torch.cuda.synchronize()
                held, beside = t0.elapsed_time(t_cur), t0.elapsed_time(t_side)
                assert beside < 0.25 * held, "the side stream waited for the held stream: %.3f of %.3f ms" % (beside, held)
    finally:
        merge._BESIDE.clear()
        merge._BESIDE.update(kept)
    del junk


def test_two_ranks_in_one_process_stream_ordered(c_oracle):
    """world_size 2 with no host drain anywhere: two threads drive one context each on streams of their own, and the
    collectives between them are carried out in stream order on a communication stream (LocalStreamWorld), with each
    rank's stream held by a sleep kernel while its chain is enqueued.  ONE job cut in two contiguous shards (and an
    empty shard on one rank), every merge flow, against one context over the whole job."""
    import threading
    import torch
    from pypore_amd import qc, synth
    from pypore_amd.merge import finalize_distributed
    from stream_collectives import LocalStreamWorld
    torch.cuda.set_device(0)
    for label, mode, kw in FLOWS:
        cfg = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=mode, tz_offset_s=-18000 * mode,
                   ref_time_s=1_528_000_000, fail_q=7.0)
        with qc.QCContext(device=0, **cfg) as whole:
            job = _device_job(torch, qc, synth, whole, 300_000, 7.0, 512, 60 if mode else 47, 90 + mode, tweak_times=mode == 1)
            keep = []
            _submit_shard(torch, whole, job, 0, job["R"], keep)
            want = whole.finalize()
        cut = job["R"] // 3
        while int(job["off"][cut]) % 16:
            cut += 1
        for cuts in ([0, cut, job["R"]], [0, job["R"], job["R"]]):
            world = LocalStreamWorld(2, torch)
            got, errors = [None, None], []

            def rank_main(r):
                try:
                    torch.cuda.set_device(0)
                    with torch.cuda.stream(torch.cuda.Stream()):
                        with qc.QCContext(device=0, **cfg) as ctx:
                            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                            mine = []
                            for rep in range(2):          # the second pass finds every buffer allocated: no host wait in it
                                ctx.reset()
                                if rep:
                                    torch.cuda._sleep(SLEEP_CYCLES)
                                _submit_shard(torch, ctx, job, cuts[r], cuts[r + 1], mine)
                                got[r] = finalize_distributed(ctx, world.rank(r), torch, **kw)
                except BaseException as e:      # the other rank would wait at the barrier for ever
                    errors.append((r, e))
                    world.bar.abort()

            th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
            for x in th:
                x.start()
            for x in th:
                x.join(120)
            assert not errors, errors
            assert not any(x.is_alive() for x in th)
            assert world.calls >= 4
            for r in range(2):
                assert_tables_equal(got[r], want, "%s, rank %d of cuts %s" % (label, r, cuts))
        if mode == 0:
            assert want["n50"] == c_oracle.n50(np.asarray(job["lens"]))


def test_single_read_channel_rule_across_ranks(qc_mod, c_oracle):
    """PPQC_COMPAT_MULTI_READ_SINGLE with a job's reads on two ranks (lib/multi_read_fast5_reader.py:64-67: `len(l) > 1`
    is the channel's read count in the whole job): ppqc_compat_dist_count + one small all-reduce SUM in front of the
    tables.  Two ranks in one process with stream-ordered collectives, every merge flow; cuts that leave single-read
    channels on either rank, channels with one read on each rank (not single), and an empty shard.  Against one context
    over the whole job, the oracle's channel_parser restatement, and ppqc_group_finalize on one GPU."""
    import threading
    import torch
    from pypore_amd.merge import finalize_distributed
    from stream_collectives import LocalStreamWorld
    qc = qc_mod
    torch.cuda.set_device(0)
    for label, mode, kw in FLOWS:
        cfg = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=mode, tz_offset_s=3600 * mode,
                   ref_time_s=1_528_000_000, fail_q=7.0, compat_flags=1)
        b = random_batch(61 + mode, 700, 150, hours=46)
        R = len(b["channel"])
        args = lambda a, e: (b["bases"][int(b["offsets"][a]):int(b["offsets"][e])], b["quals"][int(b["offsets"][a]):int(b["offsets"][e])],
                             b["offsets"][a:e + 1] - b["offsets"][a], b["channel"][a:e], b["mux"][a:e], b["time"][a:e],
                             b["fail_flag"][a:e])
        with qc.QCContext(device=0, **cfg) as whole, qc.QCContext(device=0, **dict(cfg, compat_flags=0)) as plain:
            whole.submit(*args(0, R))
            want = whole.finalize()
            plain.submit(*args(0, R))
            without = plain.finalize()
            with qc.Group([whole]) as grp:           # the library's own merge applies the rule too
                whole.reset()
                whole.submit(*args(0, R))
                assert_tables_equal(grp.finalize(), want, "group of one with the single-read-channel rule")
        w = c_oracle.per_read(b["bases"], b["quals"], b["offsets"], nthreads=4)
        assert_tables_equal(want, c_oracle.tables(w, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg))
        assert int((want["chan_reads"] == 1).sum()) > 100 and not np.array_equal(want["cmh_reads"], without["cmh_reads"])
        for cuts in ([0, R // 3, R], [0, R, R]):
            world = LocalStreamWorld(2, torch)
            got, errors = [None, None], []

            def rank_main(r):
                try:
                    torch.cuda.set_device(0)
                    with torch.cuda.stream(torch.cuda.Stream()):
                        with qc.QCContext(device=0, **cfg) as ctx:
                            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                            if cuts[r + 1] > cuts[r]:
                                ctx.submit(*args(cuts[r], cuts[r + 1]))
                            got[r] = finalize_distributed(ctx, world.rank(r), torch, **kw)
                except BaseException as e:
                    errors.append((r, e))
                    world.bar.abort()

            th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
            for x in th:
                x.start()
            for x in th:
                x.join(120)
            assert not errors, errors
            for r in range(2):
                assert_tables_equal(got[r], want, "%s with the single-read-channel rule, rank %d of cuts %s" % (label, r, cuts))


def test_mode1_corner_cases_of_the_review(qc_mod, c_oracle):
    """(1) reads tied at a channel's earliest start that lies outside the day window fail the run; (2) a read
    without a start time keeps out of the timed reads' order whatever the span of the start times (a span of 2^8k - 1
    used to tie its sort key with the latest timed read's); (3) a reference time set behind ppqc_mode1_dist_phase makes
    ppqc_finalize_tables compute the hours again instead of binning with stale ones."""
    from pypore_amd._lib import PPQCError
    qc = qc_mod
    seq = lambda n: (np.frombuffer(b"ACGT" * n, np.uint8), np.frombuffer(b"IIII" * n, np.uint8),
                     np.arange(0, 4 * n + 1, 4, dtype=np.uint64))
    cfg = dict(n_channels=4, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0, ref_time_s=10**9)
    with qc.QCContext(device=0, **cfg) as ctx:
        ctx.submit(*seq(3), np.array([1, 1, 1], np.uint32), np.array([1, 1, 1], np.uint8),
                   np.array([10**9 - 86400 * 3, 10**9 - 86400 * 3, 10**9 + 3600], np.int64))
        ctx.finalize_extremes()
        for ph in range(3):
            ctx.mode1_dist_phase(ph)
        ctx.finalize_tables()
        ctx.finalize_n50()
        with pytest.raises(PPQCError, match="day window"):
            ctx.fetch_tables()
    # (2) ten-tick bins: 0, 100, 230, 255 ticks are hours 0, 10, 23, 1 -> the last read is bumped to 25
    cfg2 = dict(n_channels=2, n_mux=4, n_time_bins=48, bin_seconds=10, time_mode=1, tz_offset_s=0, ref_time_s=0, fail_q=7.0)
    t = np.array([np.iinfo(np.int64).min, 0, 100, 230, 255], np.int64)
    ch, mux = np.ones(5, np.uint32), np.ones(5, np.uint8)
    with qc.QCContext(device=0, **cfg2) as ctx:
        ctx.submit(*seq(5), ch, mux, t)
        got = ctx.finalize()
        assert list(ctx.read_hours()) == [-1, 0, 10, 23, 25]
    b, q, o = seq(5)
    assert_tables_equal(got, c_oracle.tables(c_oracle.per_read(b, q, o, nthreads=1), ch, mux, t, None, cfg2))
    # (3)
    bb = random_batch(71, 3000, 100, n_channels=24, hours=60)
    cfg3 = dict(n_channels=24, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0,
                ref_time_s=1_528_000_000, fail_q=7.0)
    args = (bb["bases"], bb["quals"], bb["offsets"], bb["channel"], bb["mux"], bb["time"], bb["fail_flag"])
    new_ref = 1_528_000_000 - 5 * 3600
    with qc.QCContext(device=0, **dict(cfg3, ref_time_s=new_ref)) as fresh:
        fresh.submit(*args)
        want = fresh.finalize()
    with qc.QCContext(device=0, **cfg3) as ctx:
        ctx.submit(*args)
        ctx.finalize_extremes()
        for ph in range(3):
            ctx.mode1_dist_phase(ph)
        ctx.set_ref_time(new_ref)
        ctx.finalize_tables()
        ctx.finalize_n50()
        assert_tables_equal(ctx.fetch_tables(), want, "reference time set behind the phases")
