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
    assert r.returncode == 0 and "dist rehearsal ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
