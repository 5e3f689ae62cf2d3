"""Multi-GPU merge of the run-level tables: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

Reads shard naturally (every table cell is a sum / min / max of per-read contributions), so the data path has no
collective; only the small integer tables are merged -- the counterpart of gathering the per-channel results over
the multiprocessing result queue in the reference (lib/seq_routines.py:940-942) and of the sum over channels in
FastStats (lib/seq_routines.py:103-108):

  beside the per-read kernel, on a side stream (everything here needs the CSR index only, which the batches left when
  they were submitted):
  1. one all-reduce MAX on [max_len | chan_max_len | ~min_pos_len | ~chan_min_len] (a few KB; the device keeps the
     minima bitwise complemented so that they merge by MAX too)
  2. exact N50 by weighted radix select: three 2048-word histograms, each all-reduced between its histogram and its
     pick (the alternative n50="gather" all-gathers the u32 lengths and sorts them on every rank)
  behind it:
  3. all-reduce SUM on the fused u64 table buffer (1.6 MB at 512x4x48, 13.9 MB at 3000x4x72) -- one call; at these
     sizes the ring is latency-bound on xGMI, so it is not bucketed or pipelined
  (early=False, or a context that cannot tell its index-only results early: the same merges as four collectives behind
  the kernel, the first select histogram fused into the table buffer's all-reduce)

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


_SIDE = {}


def _side_stream(torch):
    """One extra stream per device for the merges that run beside the per-read kernel."""
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev)
    return _SIDE[dev]


def finalize_distributed(ctx, dist, torch, group=None, n50="select", copy=True, early=None):
    """Phased finalize of a QCContext with RCCL merges in between; returns the merged tables on every rank.

    The context is put on torch's current stream, so its kernels and the collectives (which torch orders against that
    stream with events) form one stream-ordered chain: nothing synchronises with the host before the final fetch.

    ``n50="select"`` (default): exact N50 by weighted radix select -- three rounds, each a 2048-word histogram that is
    all-reduced (the first one rides along with the table buffer), so only small tables cross xGMI and the cost does
    not grow with the number of ranks.  ``n50="gather"``: all-gather the u32 lengths and run the LSD radix sort + N50
    on every rank (the single-GPU algorithm; its sort grows with the whole job).  ``copy=False``: the tables are views
    of the context's pinned result buffer (``QCContext.fetch_tables``).

    ``early`` (default: whenever the context says it can, ``QCContext.index_early`` -- every rank must take the same
    path, and does when all ranks drive their contexts through the same calls): the extremes all-reduce and the three
    select rounds depend on the CSR index only, which the batches left when they were submitted; they are issued on a
    side stream and run WHILE the per-read kernel does, so that one collective (the table SUM) is left behind it."""
    handle = torch.cuda.current_stream().cuda_stream
    if ctx.external_stream != handle:
        ctx.set_stream(handle)
    if early is None:
        early = n50 == "select" and ctx.index_early()
    if early:
        side = _side_stream(torch)
        ctx.stream_wait_index(side.cuda_stream)
        views, _ = ctx.device_tables()
        with torch.cuda.stream(side):
            t_maxmin = torch.as_tensor(views["maxmin"], device="cuda")
            dist.all_reduce(t_maxmin, op=dist.ReduceOp.MAX, group=group)
            t_sel = torch.as_tensor(views["sel"], device="cuda")
            for r in range(3):
                ctx.n50_select_early(r, 0, side.cuda_stream)
                dist.all_reduce(t_sel, op=dist.ReduceOp.SUM, group=group)
                ctx.n50_select_early(r, 1, side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        ctx.finalize_extremes()
        ctx.finalize_tables()                    # uses the merged max_len; the N50 is final already
        t_sum = torch.as_tensor(views["sum"], device="cuda")
        dist.all_reduce(t_sum, op=dist.ReduceOp.SUM, group=group)
        return ctx.fetch_tables(copy=copy)
    ctx.finalize_extremes()
    views, _ = ctx.device_tables()
    # the device keeps the minima complemented, so maxima and minima merge in ONE all-reduce(MAX)
    t_maxmin = torch.as_tensor(views["maxmin"], device="cuda")
    dist.all_reduce(t_maxmin, op=dist.ReduceOp.MAX, group=group)
    ctx.finalize_tables()                        # length histogram bins use the merged max_len
    if n50 == "select":
        t_sum_sel = torch.as_tensor(views["sum_sel"], device="cuda")
        t_sel = torch.as_tensor(views["sel"], device="cuda")
        ctx.n50_select_hist(0)
        dist.all_reduce(t_sum_sel, op=dist.ReduceOp.SUM, group=group)      # tables + first select histogram
        ctx.n50_select_pick(0)
        for r in (1, 2):
            ctx.n50_select_hist(r)
            dist.all_reduce(t_sel, op=dist.ReduceOp.SUM, group=group)
            ctx.n50_select_pick(r)
        return ctx.fetch_tables(copy=copy)
    t_sum = torch.as_tensor(views["sum"], device="cuda")
    dist.all_reduce(t_sum, op=dist.ReduceOp.SUM, group=group)
    n_local = views["lengths"].__cuda_array_interface__["shape"][0]
    if n_local:
        lens = torch.as_tensor(views["lengths"], device="cuda")
    else:
        lens = torch.zeros(0, dtype=torch.int32, device="cuda")
    all_lens = gather_lengths(lens, dist, torch, group)
    ctx.finalize_n50(all_lens.data_ptr(), all_lens.numel())
    tables = ctx.fetch_tables(copy=copy)         # synchronises: all_lens may be released afterwards
    return tables


SELECT_ROUNDS = ((21, 11), (10, 11), (0, 10))    # (shift, bits) of ppqc_n50_select_hist / _pick


def n50_select_reference(local_lengths, all_reduce_sum):
    """The arithmetic of the select path on host arrays, for tests: ``local_lengths`` is this rank's uint32 vector,
    ``all_reduce_sum(np_int64_array)`` merges a histogram (or a scalar) across ranks in place."""
    import numpy as np
    lens = np.asarray(local_lengths, dtype=np.uint32)
    total = np.array([int(lens.sum(dtype=np.uint64))], dtype=np.int64)
    all_reduce_sum(total)
    T = int(total[0])
    prefix, above = 0, 0
    for shift, bits in SELECT_ROUNDS:
        hi = shift + bits
        m = np.ones(len(lens), bool) if hi >= 32 else ((lens >> np.uint32(hi)) == (prefix >> hi))
        m &= lens > 0
        digit = ((lens[m] >> np.uint32(shift)) & np.uint32((1 << bits) - 1)).astype(np.int64)
        w = np.zeros(1 << bits, dtype=np.int64)
        np.add.at(w, digit, lens[m].astype(np.int64))
        all_reduce_sum(w)
        if T == 0:
            continue
        after = above
        for d in range((1 << bits) - 1, -1, -1):
            suf = after + int(w[d])
            if 2 * suf >= T and 2 * after < T:
                prefix |= d << shift
                above = after
                break
            after = suf
    return prefix if T else 0


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
