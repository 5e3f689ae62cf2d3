"""Multi-GPU merge of the run-level tables: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

Reads shard naturally (every table cell is a sum / min / max of per-read contributions), so the data path has no
collective; only the small integer tables are merged -- the counterpart of gathering the per-channel results over
the multiprocessing result queue in the reference (lib/seq_routines.py:940-942) and of the sum over channels in
FastStats (lib/seq_routines.py:103-108):

  1. all-reduce MAX on [max_len | chan_max_len], MIN on [min_pos_len | chan_min_len]   (a few KB)
  2. all-reduce SUM on the fused u64 table buffer (1.6 MB at 512x4x48, 13.9 MB at 3000x4x72) -- one call; at these
     sizes the ring is latency-bound on xGMI, so it is not bucketed or pipelined
  3. all-gather of the u32 read lengths (4 MB per 1e6 reads), then every rank runs the same radix sort + N50

time_mode 1 (the reference's per-channel hour-wrap quirk) needs all reads of a channel on one rank: shard by channel.
"""
from __future__ import annotations


def allreduce_table_tensors(sum_t, max_t, min_t, dist, group=None):
    """In-place merge of the three table tensors (int64).  Works on any backend."""
    dist.all_reduce(max_t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(min_t, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(sum_t, op=dist.ReduceOp.SUM, group=group)


def gather_lengths(lengths_t, dist, torch, group=None):
    """All-gather of per-rank length vectors of different sizes (zero padding does not change N50)."""
    world = dist.get_world_size(group)
    n = torch.tensor([lengths_t.numel()], dtype=torch.int64, device=lengths_t.device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    nmax = int(n.item())
    padded = torch.zeros(nmax, dtype=lengths_t.dtype, device=lengths_t.device)
    padded[: lengths_t.numel()] = lengths_t
    out = torch.empty(nmax * world, dtype=lengths_t.dtype, device=lengths_t.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return out


def finalize_distributed(ctx, dist, torch, group=None):
    """Phased finalize of a QCContext with RCCL merges in between; returns the merged tables on every rank."""
    ctx.finalize_extremes()
    views, _ = ctx.device_tables()
    t_max = torch.as_tensor(views["max"], device="cuda")
    t_min = torch.as_tensor(views["min"], device="cuda")
    t_sum = torch.as_tensor(views["sum"], device="cuda")
    ctx.sync()                                   # library stream -> host; torch's stream picks up from here
    dist.all_reduce(t_max, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(t_min, op=dist.ReduceOp.MIN, group=group)
    torch.cuda.current_stream().synchronize()
    ctx.finalize_tables()                        # length histogram bins use the merged max_len
    ctx.sync()
    dist.all_reduce(t_sum, op=dist.ReduceOp.SUM, group=group)
    n_local = views["lengths"].__cuda_array_interface__["shape"][0]
    if n_local:
        lens = torch.as_tensor(views["lengths"], device="cuda")
    else:
        lens = torch.zeros(0, dtype=torch.int32, device="cuda")
    all_lens = gather_lengths(lens, dist, torch, group)
    torch.cuda.current_stream().synchronize()
    ctx.finalize_n50(all_lens.data_ptr(), all_lens.numel())
    return ctx.fetch_tables()


def merge_tables_host(tables_list):
    """Reference semantics of the merge on host dicts (as returned by QCContext.fetch_tables / the oracle):
    used by the tests to state 'N shards merged == 1 shard over the concatenation'."""
    import numpy as np
    sum_keys = ("totals", "len_hist_log2", "q_hist", "gc_hist", "hour_reads", "hour_bases", "chan_reads",
                "chan_bases", "chan_fail", "cmh_reads", "cmh_bases")
    out = {k: sum(np.asarray(t[k]).astype(np.int64) for t in tables_list) for k in sum_keys}
    out["chan_max_len"] = np.maximum.reduce([np.asarray(t["chan_max_len"]).astype(np.int64) for t in tables_list])
    out["chan_min_len"] = np.minimum.reduce([np.asarray(t["chan_min_len"]).astype(np.int64) for t in tables_list])
    out["max_len"] = max(int(t["max_len"]) for t in tables_list)
    out["min_pos_len"] = min(int(t["min_pos_len"]) for t in tables_list)
    return out
