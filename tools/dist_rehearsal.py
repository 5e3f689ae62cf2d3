"""Two ranks sharing ONE GPU over gloo (RCCL refuses two ranks on one device): rehearses the control flow of
finalize_distributed with world_size 2 -- the early merge on a side stream, the phased merge, an empty shard on one
rank -- and checks every rank's merged tables against the host merge of the single-rank tables and the oracle's N50
over all lengths.  GPU box only.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/dist_rehearsal.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
from pypore_amd import qc
from pypore_amd.merge import finalize_distributed, merge_tables_host
from oracle import c_oracle as co
from util import random_batch, assert_tables_equal

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
dist.barrier()
if rank == 0:
    print("dist rehearsal ok: early, phased, gather, empty shard; n50", want_n50)
dist.destroy_process_group()
