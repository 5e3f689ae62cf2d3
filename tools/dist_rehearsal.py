"""Two ranks sharing ONE GPU over gloo (RCCL refuses two ranks on one device): rehearses the control flow of
finalize_distributed with world_size 2 -- the early merge on a side stream, the phased merge, an empty shard on one
rank -- and checks every rank's merged tables against the host merge of the single-rank tables and the oracle's N50
over all lengths.  GPU box only.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/dist_rehearsal.py"""
import os, sys, faulthandler
import numpy as np
faulthandler.dump_traceback_later(int(os.environ.get("PPQC_REHEARSAL_WATCHDOG", "120")), exit=True)   # a hang names itself
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
from pypore_amd import qc
from pypore_amd.merge import finalize_distributed, merge_tables_host
from oracle import c_oracle as co
from util import random_batch, assert_tables_equal

if os.environ.get("PPQC_REHEARSAL_SYNC"):   # diagnosis of a hang: synchronise behind every phase, so the watchdog names it
    for _name in ("finalize_extremes", "finalize_tables", "mode1_dist_phase", "n50_select_hist", "n50_select_pick",
                  "n50_select_early", "finalize_n50"):
        def _wrap(f):
            def g(self, *a, **k):
                r = f(self, *a, **k)
                qc.L.check(self._lib.ppqc_sync(self._ctx))
                torch.cuda.synchronize()
                return r
            return g
        setattr(qc.QCContext, _name, _wrap(getattr(qc.QCContext, _name)))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = dict(n_channels=512, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=0, tz_offset_s=0,
           ref_time_s=1_528_000_000, fail_q=7.0)
batches = [random_batch(100 + r, 3000 + 2000 * r, 700) for r in range(world)]
b = batches[rank]


def submit(ctx, bb):
    ctx.submit(bb["bases"], bb["quals"], bb["offsets"], bb["channel"], bb["mux"], bb["time"], bb["fail_flag"])


singles = []
for bb in batches:                       # every rank computes all single-shard tables (reference for the merge)
    with qc.QCContext(device=0, **cfg) as c1:
        submit(c1, bb)
        singles.append(c1.finalize())
want = merge_tables_host(singles)
all_lens = np.concatenate([np.where((bb["channel"] >= 1) & (bb["channel"] <= 512) & (bb["fail_flag"] == 0),
                                    np.diff(bb["offsets"].astype(np.int64)), 0) for bb in batches])
want_n50 = co.n50(all_lens.astype(np.uint32))
with qc.QCContext(device=0, **cfg) as ctx:
    for label, kw in (("early", {}), ("phased", dict(early=False)), ("gather", dict(n50="gather"))):
        ctx.reset()
        submit(ctx, b)
        got = finalize_distributed(ctx, dist, torch, **kw)
        for k in want:
            assert np.array_equal(np.asarray(got[k]).astype(np.int64), np.asarray(want[k]).astype(np.int64)), (label, k)
        assert got["n50"] == want_n50, (label, got["n50"], want_n50)
    # rank 1 holds an empty shard: both ranks must still take the early path and agree
    ctx.reset()
    if rank == 0:
        submit(ctx, b)
    assert ctx.index_early()
    got = finalize_distributed(ctx, dist, torch)
    assert_tables_equal(got, singles[0], "empty shard")
# time_mode 1 with every channel's reads on both ranks: ONE batch cut in two contiguous halves (channels are random over
# the reads), merged tables against one context over the whole batch (the sort-based hours) -- and an empty shard
cfg1 = dict(cfg, time_mode=1, tz_offset_s=-18000)
bb = random_batch(300, 9000, 400, hours=60)
R = len(bb["channel"])
with qc.QCContext(device=0, **cfg1) as c1:
    submit(c1, bb)
    whole = c1.finalize()
for cut in ([0, R // 3, R], [0, R, R]):
    a, e = cut[rank], cut[rank + 1]
    o = bb["offsets"][a:e + 1]
    with qc.QCContext(device=0, **cfg1) as ctx:
        if e > a:
            ctx.submit(bb["bases"][int(o[0]):int(o[-1])], bb["quals"][int(o[0]):int(o[-1])], o - o[0], bb["channel"][a:e],
                       bb["mux"][a:e], bb["time"][a:e], bb["fail_flag"][a:e])
        got = finalize_distributed(ctx, dist, torch)
    assert_tables_equal(got, whole, "time_mode 1 across ranks, cut %s" % cut)
dist.barrier()
if rank == 0:
    print("dist rehearsal ok: early, phased, gather, empty shard, time_mode 1 across ranks; n50", want_n50)
dist.destroy_process_group()
