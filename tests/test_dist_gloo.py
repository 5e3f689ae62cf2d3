"""N > 1 path on CPU: two processes, gloo backend.  'N shards merged == 1 shard over the concatenation' for every
mergeable table, the extremes and N50 (all-gather of lengths), using the same merge routines bench.py uses with RCCL."""
import os
import socket

import numpy as np
import pytest

from util import random_batch


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
    from oracle import pypore_oracle as po, c_oracle as co
    from pypore_amd import merge
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = dict(n_channels=24, n_mux=4, n_time_bins=72, bin_seconds=3600, time_mode=0, tz_offset_s=0,
                   ref_time_s=1_528_000_000, fail_q=7.0)
        b = random_batch(77, 3001, 150, n_channels=24, hours=75)
        R = len(b["channel"])
        cut = [0, R // 3, R][rank:rank + 2]       # uneven contiguous shards
        a, e = cut
        o = b["offsets"][a:e + 1]
        st = co.per_read(b["bases"][int(o[0]):int(o[-1])], b["quals"][int(o[0]):int(o[-1])], o - o[0], 2)
        meta = (b["channel"][a:e], b["mux"][a:e], b["time"][a:e], b["fail_flag"][a:e])
        # phase 1: local extremes -> all-reduce MAX / MIN
        loc = co.tables(st, *meta, cfg)
        t_max = torch.from_numpy(np.concatenate(([loc["max_len"]], loc["chan_max_len"])).astype(np.int64))
        t_min = torch.from_numpy(np.concatenate(([loc["min_pos_len"]], loc["chan_min_len"])).astype(np.int64))
        keys = ("totals", "len_hist_log2", "q_hist", "gc_hist", "hour_reads", "hour_bases", "chan_reads", "chan_bases",
                "chan_fail", "cmh_reads", "cmh_bases")
        t_sum = torch.from_numpy(np.concatenate([np.asarray(loc[k]).reshape(-1) for k in keys]).astype(np.int64))
        # the device layout [max | ~min] merges in ONE all-reduce(MAX) (finalize_distributed); same result as MAX + MIN
        t_maxmin = torch.cat([t_max, ~t_min])
        dist.all_reduce(t_maxmin, op=dist.ReduceOp.MAX)
        merge.allreduce_table_tensors(t_sum, t_max, t_min, dist)
        assert torch.equal(t_maxmin[: t_max.numel()], t_max) and torch.equal(~t_maxmin[t_max.numel():], t_min)
        lens = np.where((meta[0] >= 1) & (meta[0] <= 24) & (meta[3] == 0), st["length"], 0).astype(np.int32)
        all_lens = merge.gather_lengths(torch.from_numpy(lens), dist, torch)
        # reference: one shard over the concatenation
        full = co.per_read(b["bases"], b["quals"], b["offsets"], 2)
        want = co.tables(full, b["channel"], b["mux"], b["time"], b["fail_flag"], cfg)
        wsum = np.concatenate([np.asarray(want[k]).reshape(-1) for k in keys]).astype(np.int64)
        assert np.array_equal(t_sum.numpy(), wsum)
        assert int(t_max[0]) == want["max_len"] and np.array_equal(t_max[1:].numpy(), want["chan_max_len"])
        assert int(t_min[0]) == want["min_pos_len"] and np.array_equal(t_min[1:].numpy(), want["chan_min_len"])
        assert all_lens.numel() == 2 * max(R // 3, R - R // 3)
        assert co.n50(all_lens.numpy().astype(np.uint32)) == want["n50"]
        # the select path (bench.py --gpus N): same N50 from three all-reduced 2048-word histograms, no gather
        def ar(a):
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert merge.n50_select_reference(lens.astype(np.uint32), ar) == want["n50"]
        for case in ([[5], []], [[2, 2, 2, 3], [3, 4, 8, 8]], [[], []], [[1 << 31, 7], [(1 << 31) + 5, 1 << 22]],
                     [[2_000_000] * 3, list(range(1, 3000))]):
            mine = np.asarray(case[rank], dtype=np.uint32)
            assert merge.n50_select_reference(mine, ar) == po.n50([int(x) for x in case[0] + case[1]]), case
        # PPQC_COMPAT_MULTI_READ_SINGLE across ranks (merge._merge_compat_counts + compat_single_kernel): per-channel read
        # counts all-reduced, the rule applied by the rank that holds a single-read channel's read, tables summed
        cfgc = dict(n_channels=40, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0, tz_offset_s=0,
                    ref_time_s=1_528_000_000, fail_q=7.0)
        bc = random_batch(78, 60, 60, edge=False, n_channels=40, hours=47)
        Rc = len(bc["channel"])
        for ca, ce in ([0, Rc // 2, Rc][rank:rank + 2], [0, Rc, Rc][rank:rank + 2]):     # halves; an empty shard on rank 1
            oc = bc["offsets"][ca:ce + 1]
            stc = po.per_read_stats(bc["bases"][int(oc[0]):int(oc[-1])], bc["quals"][int(oc[0]):int(oc[-1])], oc - oc[0])
            mc = (bc["channel"][ca:ce], bc["mux"][ca:ce], bc["time"][ca:ce], bc["fail_flag"][ca:ce])
            lc = po.build_tables(stc, *mc, cfg=cfgc)
            cnt = torch.from_numpy(np.asarray(lc["chan_reads"], np.int64).copy())
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            lenc = np.where(mc[3] != 0, 0, stc["length"])
            merge.compat_single_reference(lc, mc[0], mc[1], lenc, lc["fail"], lc["hour"], cnt.numpy(), cfgc)
            fullc = po.build_tables(po.per_read_stats(bc["bases"], bc["quals"], bc["offsets"]), bc["channel"], bc["mux"],
                                    bc["time"], bc["fail_flag"], cfg=dict(cfgc, compat_flags=1))
            assert int((np.asarray(fullc["chan_reads"]) == 1).sum()) >= 8
            for k in ("hour_reads", "hour_bases", "chan_reads", "chan_fail", "cmh_reads", "cmh_bases"):
                tk = torch.from_numpy(np.ascontiguousarray(lc[k], dtype=np.int64).copy())
                dist.all_reduce(tk, op=dist.ReduceOp.SUM)
                assert np.array_equal(tk.numpy(), np.asarray(fullc[k])), ("compat across ranks", k)
        hm = merge.merge_tables_host([loc, loc])
        assert np.array_equal(hm["chan_reads"], 2 * np.asarray(loc["chan_reads"]))
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_merge_equals_single_shard(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
