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
            merged = finalize_distributed(ctx, dist, torch)                  # N50 by radix select (default)
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
