"""time_mode 1 across ranks (pypore_amd/merge.py, ppqc_mode1_dist_phase): the hour rule of lib/fast5_reader.py:117-126
from three small per-channel tables instead of a channel's reads in one place.  CPU: the host statement of the phases
(merge.mode1_hours_reference) against the oracle's sequential rule, on one rank and on two gloo ranks with the reads of
every channel split between them."""
import os
import socket

import numpy as np
import pytest

from oracle import pypore_oracle as po
from pypore_amd import merge

I64_MIN = np.iinfo(np.int64).min


def oracle_hours(channel, t, cfg):
    """The sequential rule, channel by channel (oracle/pypore_oracle.py:_hour_wrap over the stable time order)."""
    nch, nb = cfg["n_channels"], cfg["n_time_bins"]
    hours = np.full(len(t), -1, np.int64)
    timed = t != I64_MIN
    sec = np.where(timed, t, 0) - cfg["ref_time_s"] + cfg["tz_offset_s"]
    hod = np.mod(np.floor_divide(sec, cfg["bin_seconds"]), 24)
    for c in range(1, nch + 1):
        idx = np.nonzero((channel == c) & timed)[0]
        idx = idx[np.argsort(t[idx], kind="stable")]
        hours[idx] = po._hour_wrap([int(h) for h in hod[idx]])
    hours[hours >= nb] = -1
    return hours.astype(np.int32)


def make_case(seed, n, nch=12, span_h=60, ref=1_528_000_000, tz=0, cluster=False):
    rng = np.random.default_rng(seed)
    ch = rng.integers(0, nch + 2, n).astype(np.int64)          # 0 and nch + 1: bad channels
    if cluster:   # few distinct times: ties at the earliest start, channels with one or two reads, lone days
        t = ref + rng.choice(np.array([0, 0, 3600 * 23, 3600 * 25, 3600 * 26, 3600 * 49, 3600 * 47 + 5]), n)
    else:
        t = ref + rng.integers(-3600 * 3 if tz else 0, 3600 * span_h, n)
    t = t.astype(np.int64)
    t[rng.random(n) < 0.03] = I64_MIN                            # reads without a start time
    cfg = dict(n_channels=nch, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=tz, ref_time_s=ref)
    return ch, t, cfg


@pytest.mark.parametrize("seed,n,kw", [(1, 4000, {}), (2, 300, {}), (3, 60, dict(cluster=True)), (4, 25, dict(cluster=True)),
                                       (5, 3000, dict(tz=-18000)), (6, 3000, dict(tz=7200, span_h=100)),
                                       (7, 12, dict(cluster=True)), (8, 2000, dict(span_h=30))])
def test_one_rank_equals_sequential_rule(seed, n, kw):
    ch, t, cfg = make_case(seed, n, **kw)
    ident = lambda a: None
    got, oob = merge.mode1_hours_reference(ch, t, cfg, ident, ident)
    assert oob == 0
    assert np.array_equal(got, oracle_hours(ch, t, cfg))


def test_hand_cases():
    cfg = dict(n_channels=1, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0, ref_time_s=0)
    ident = lambda a: None
    H = 3600

    def run(times):
        t = np.asarray(times, np.int64)
        got, _ = merge.mode1_hours_reference(np.ones(len(t), np.int64), t, cfg, ident, ident)
        assert np.array_equal(got, oracle_hours(np.ones(len(t), np.int64), t, cfg)), times
        return list(got)
    # the descent between index 0 and 1 is never looked at (e > 1): 23h, 1h (next day), 2h stays unbumped
    assert run([23 * H, 25 * H, 26 * H]) == [23, 1, 2]
    # ... but one between index 1 and 2 is
    assert run([22 * H, 23 * H, 25 * H, 26 * H]) == [22, 23, 25, 26]
    # two reads at the earliest second: one of them is index 1 and takes part
    assert run([23 * H, 23 * H, 25 * H]) == [23, 23, 25]
    # a later day whose hour is not below the earlier maximum is not a descent: 1h, 5h next day
    assert run([0, 1 * H, 29 * H, 30 * H]) == [0, 1, 5, 6]
    # bumped once, bumped for good (third day keeps + 24 and leaves the 48 bins)
    assert run([0, 10 * H, 25 * H, 50 * H]) == [0, 10, 25, 26]


def test_reads_outside_the_day_window_are_counted():
    cfg = dict(n_channels=2, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0, ref_time_s=10**9)
    ident = lambda a: None
    t = np.asarray([10**9, 10**9 + 5, 10**9 + 86400 * 70, 10**9 - 86400 * 3], np.int64)
    _, oob = merge.mode1_hours_reference(np.asarray([1, 1, 1, 2]), t, cfg, ident, ident)
    assert oob == 1        # the read 70 days on; the lone early read of channel 2 is that channel's first and has no day
    # two reads tied at a channel's earliest start, outside the window: one of them belongs to S, whose day tables cannot
    # hold it -- counted, so the run fails instead of dropping its hour from the prefix maximum
    t = np.asarray([10**9 - 86400 * 3, 10**9 - 86400 * 3, 10**9 + 3600], np.int64)
    _, oob = merge.mode1_hours_reference(np.asarray([1, 1, 1]), t, cfg, ident, ident)
    assert oob == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def ar(op):
            def f(a):
                tt = torch.from_numpy(a)
                dist.all_reduce(tt, op=op)
            return f
        for seed, n, kw in [(11, 5000, {}), (12, 80, dict(cluster=True)), (13, 2000, dict(tz=-18000)), (14, 9, dict(cluster=True))]:
            ch, t, cfg = make_case(seed, n, **kw)
            want = oracle_hours(ch, t, cfg)
            # every channel's reads on both ranks: odd / even reads, and (second pass) an empty shard on rank 1
            for mine in (np.arange(n) % world == rank, np.full(n, rank == 0)):
                got, oob = merge.mode1_hours_reference(ch[mine], t[mine], cfg, ar(dist.ReduceOp.MAX), ar(dist.ReduceOp.SUM))
                assert oob == 0
                assert np.array_equal(got, want[mine]), (seed, rank)
        assert merge.shard_files(list(range(7)), rank, world) == [[0, 1, 2], [3, 4, 5, 6]][rank]
        assert merge.allreduce_min_int(100 - rank, dist, torch, device="cpu") == 99
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_sharing_every_channel(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
