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
  3. all-reduce SUM on the (channel, mux, hour) cells, which are index-only too and all but a sliver of the table buffer
     (1.5 MB at 512x4x48, 13.8 MB at 3000x4x72) -- one call; at these sizes the ring is latency-bound on xGMI, so it is
     not bucketed or pipelined -- and their copy to the host
  behind it:
  4. all-reduce SUM on the rest of the table buffer (histograms, per-hour and per-channel tables: a few thousand words)
  (early=False, or a context that cannot tell its index-only results early: the same merges as four collectives behind
  the kernel, the first select histogram fused into the table buffer's all-reduce)

time_mode 1 (the reference's per-channel hour-wrap quirk, lib/fast5_reader.py:117-126) is defined on a channel's reads
in start-time order, but it needs no reads in one place: the ranks agree on three small per-channel tables instead (the
channel's earliest start, the reads tied at it, per day the earliest and latest start of the others; three more small
all-reduces, ppqc_mode1_dist_phase) and every rank then bins its own reads.  So the CLI shards FILES, whatever channels
they hold (``shard_files``).  ``mode1_hours_reference`` is the same arithmetic on host arrays, for the tests.
"""
from __future__ import annotations


def allreduce_table_tensors(sum_t, max_t, min_t, dist, group=None):
    """In-place merge of the three table tensors (int64).  Works on any backend."""
    dist.all_reduce(max_t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(min_t, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(sum_t, op=dist.ReduceOp.SUM, group=group)


def _gloo(dist, group):
    return dist.get_backend(group) == "gloo"


def gather_lengths(lengths_t, dist, torch, group=None):
    """All-gather of per-rank length vectors of different sizes (zero padding does not change N50)."""
    world = dist.get_world_size(group)
    dev = lengths_t.device
    if lengths_t.is_cuda and _gloo(dist, group):
        lengths_t = lengths_t.cpu()                    # gloo only ever sees host memory: see _all_reduce
    n = torch.tensor([lengths_t.numel()], dtype=torch.int64, device=lengths_t.device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    nmax = int(n.item())
    padded = torch.zeros(nmax, dtype=lengths_t.dtype, device=lengths_t.device)
    padded[: lengths_t.numel()] = lengths_t
    out = torch.empty(nmax * world, dtype=lengths_t.dtype, device=lengths_t.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return out.to(dev)


def _all_reduce(dist, torch, t, op, group=None):
    """dist.all_reduce of a device tensor, in place.

    RCCL ("nccl") enqueues the collective on the issuing stream's order and nothing waits on the host.

    gloo (the CPU tests and the rehearsals with several ranks on one GPU) has no device path of its own: handed a device
    tensor, ProcessGroupGloo copies it to pinned memory on a high-priority pool stream that waits for the issuing stream,
    and a worker thread blocks in that pool stream's synchronize() before the host exchange.  With two processes sharing
    one GPU that wait did not return when kernels were still queued in front of it (five recorded runs, both ranks inside
    work.wait() of one and the same collective, at two collectives whose producers share nothing -- one of them a kernel
    on the very stream the collective was issued on, so no event can have been missing; a missing event gives wrong
    values, not a wait without end, and the library enqueues no kernel that waits for anything: DESIGN section 5).  So
    gloo is never handed device memory here: the tensor is copied to the host in stream order (which waits for exactly
    its producers), reduced there, and copied back.  Every gloo collective of this package goes through this function,
    ``gather_lengths`` or ``allreduce_min_int``, which do the same."""
    if t.is_cuda and _gloo(dist, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return
    dist.all_reduce(t, op=op, group=group)


def _merge_compat_counts(ctx, dist, torch, group):
    """PPQC_COMPAT_MULTI_READ_SINGLE (channel_parser's ``len(l) > 1``, lib/multi_read_fast5_reader.py:64): the rule looks at
    a channel's read count in the whole job, so the ranks add their counts up before the tables are built."""
    from ._lib import COMPAT_MULTI_READ_SINGLE
    if ctx.cfg.compat_flags & COMPAT_MULTI_READ_SINGLE:
        _all_reduce(dist, torch, torch.as_tensor(ctx.compat_dist_count(), device="cuda"), dist.ReduceOp.SUM, group)


EARLY_DEFAULT = True    # finalize_distributed(early=None): beside the per-read kernel whenever the context can
CELLS_BESIDE = True     # ... and in the flow behind the kernel, the cells alone still go beside it
_BESIDE = {}


def _beside_stream(torch):
    """A stream whose work really runs beside the current stream's.  HIP maps its streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, four by default), and two streams on one queue run in submission order whatever their events
    allow: a side stream that shared the per-read stream's queue ran its whole chain BEHIND the per-read kernel (six
    collectives in the step's tail instead of one), and the 13.8 MB copy of config 3's cells sat in front of the select
    rounds it was meant to run beside.  Which streams share depends on everything the process has created, so it is
    measured, once per device, thread and current stream: a short sleep on the current stream, an event on a candidate;
    the first of a few fresh streams whose event does not wait for the sleep is kept.  On such a stream kernels launched
    beside ``tile_kernel`` start at once (17-40 us each); no priority is asked for (a high-priority stream gave the same
    timeline, and stalled the suite: DESIGN section 5)."""
    import threading
    cur = torch.cuda.current_stream()
    key = (torch.cuda.current_device(), threading.get_ident(), cur.cuda_stream)
    if key in _BESIDE:
        return _BESIDE[key]
    chosen = None
    for _ in range(6):
        cand = torch.cuda.Stream(device=key[0])
        t0, t_cur, t_cand = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        cur.synchronize()
        cand.synchronize()
        t0.record(cur)
        torch.cuda._sleep(400_000)             # on the current stream: a fraction of a millisecond to a few
        t_cur.record(cur)
        cand.wait_event(t0)
        t_cand.record(cand)
        cur.synchronize()
        cand.synchronize()
        if chosen is None:
            chosen = cand
        if t0.elapsed_time(t_cand) < 0.5 * t0.elapsed_time(t_cur):
            chosen = cand
            break
    _BESIDE[key] = chosen
    return chosen


def _cells_to_host(ctx, torch):
    """The (channel, mux, hour) cells -- all but a sliver of the table buffer, 13.8 MB at 3000 x 4 x 72 -- to the host result
    buffer on the side stream, as soon as the current stream has merged them: beside what is still to come on that stream
    (select rounds, a gather and sort), and the fetch at the end copies the small tables only.  (Not under
    PPQC_COMPAT_MULTI_READ_SINGLE: QCContext.cells_early refuses there.)"""
    from ._lib import COMPAT_MULTI_READ_SINGLE
    if ctx.cfg.compat_flags & COMPAT_MULTI_READ_SINGLE:
        return
    cs = _beside_stream(torch)
    cs.wait_stream(torch.cuda.current_stream())
    ctx.cells_early(1, cs.cuda_stream)


def _tensor(ctx, torch, views, name):
    """torch tensor over one of the context's device tables, made once per context and table (the buffers do not move;
    a finalize per step would otherwise build a dozen tensors and views every time)."""
    cache = ctx.__dict__.setdefault("_dist_tensors", {})
    ptr = views[name].__cuda_array_interface__["data"][0]
    n = views[name].__cuda_array_interface__["shape"][0]
    hit = cache.get(name)
    if hit is None or hit[0] != (ptr, n):
        hit = cache[name] = ((ptr, n), torch.as_tensor(views[name], device="cuda"))
    return hit[1]


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
    side stream, BESIDE the per-read kernel -- and so are the (channel, mux, hour) cells, all but a sliver of the table
    buffer, their all-reduce and their copy to the host (``QCContext.cells_early``; not under
    PPQC_COMPAT_MULTI_READ_SINGLE, which rewrites cells behind the tables) -- so that one collective over a few thousand
    words (the histograms, the per-hour and per-channel tables) is left behind it.  What that takes of the side stream
    -- a hardware queue it does not share with the per-read stream, found by a probe, and no priority -- is in
    ``_beside_stream`` and in DESIGN section 5, "Beside the kernel or behind it".  ``early=False``: four collectives
    behind the kernel, with the cells alone still merged and copied beside it where the context can say so."""
    handle = torch.cuda.current_stream().cuda_stream
    if ctx.external_stream != handle:
        ctx.set_stream(handle)
    if ctx.cfg.time_mode == 1:
        return _finalize_mode1(ctx, dist, torch, group, copy)
    if early is None:
        early = EARLY_DEFAULT and n50 == "select" and ctx.index_early()
    if early:
        side = _beside_stream(torch)
        ctx.stream_wait_index(side.cuda_stream)
        views, _ = ctx.device_tables()
        from ._lib import COMPAT_MULTI_READ_SINGLE
        cells = not (ctx.cfg.compat_flags & COMPAT_MULTI_READ_SINGLE)
        with torch.cuda.stream(side):
            t_maxmin = _tensor(ctx, torch, views, "maxmin")
            _all_reduce(dist, torch, t_maxmin, dist.ReduceOp.MAX, group)
            if cells:
                ctx.cells_early(0, side.cuda_stream)             # unpacked (or, past a packed cell's range, computed) behind side
                _all_reduce(dist, torch, _tensor(ctx, torch, views, "cmh"), dist.ReduceOp.SUM, group)
                ctx.cells_early(1, side.cuda_stream)             # the merged cells to the host result buffer, on side
            t_sel = _tensor(ctx, torch, views, "sel")
            for r in range(3):
                ctx.n50_select_early(r, 0, side.cuda_stream)
                _all_reduce(dist, torch, t_sel, dist.ReduceOp.SUM, group)
                ctx.n50_select_early(r, 1, side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        ctx.finalize_extremes()
        _merge_compat_counts(ctx, dist, torch, group)
        ctx.finalize_tables()                    # uses the merged max_len; the N50 is final already
        t_sum = _tensor(ctx, torch, views, "small" if cells else "sum")
        _all_reduce(dist, torch, t_sum, dist.ReduceOp.SUM, group)
        return ctx.fetch_tables(copy=copy)
    from ._lib import COMPAT_MULTI_READ_SINGLE
    cells_beside = (CELLS_BESIDE and n50 == "select" and not (ctx.cfg.compat_flags & COMPAT_MULTI_READ_SINGLE)
                    and ctx.index_early())
    views, _ = ctx.device_tables()
    if cells_beside:
        # the cells alone beside the per-read kernel (three launches on a stream of another hardware queue: slow there,
        # but done long before that kernel is): unpack, all-reduce, copy to the host
        cs = _beside_stream(torch)
        ctx.stream_wait_index(cs.cuda_stream)
        with torch.cuda.stream(cs):
            ctx.cells_early(0, cs.cuda_stream)
            _all_reduce(dist, torch, _tensor(ctx, torch, views, "cmh"), dist.ReduceOp.SUM, group)
            ctx.cells_early(1, cs.cuda_stream)
    ctx.finalize_extremes()
    # the device keeps the minima complemented, so maxima and minima merge in ONE all-reduce(MAX)
    t_maxmin = _tensor(ctx, torch, views, "maxmin")
    _all_reduce(dist, torch, t_maxmin, dist.ReduceOp.MAX, group)
    _merge_compat_counts(ctx, dist, torch, group)
    ctx.finalize_tables()                        # length histogram bins use the merged max_len
    if n50 == "select":
        t_sum_sel = _tensor(ctx, torch, views, "sum_sel")
        t_sel = _tensor(ctx, torch, views, "sel")
        ctx.n50_select_hist(0)
        if cells_beside:
            _all_reduce(dist, torch, _tensor(ctx, torch, views, "small"), dist.ReduceOp.SUM, group)
            _all_reduce(dist, torch, t_sel, dist.ReduceOp.SUM, group)
        else:
            _all_reduce(dist, torch, t_sum_sel, dist.ReduceOp.SUM, group)  # tables + first select histogram
            _cells_to_host(ctx, torch)           # the merged cells are final: their copy beside the last two rounds
        ctx.n50_select_pick(0)
        for r in (1, 2):
            ctx.n50_select_hist(r)
            _all_reduce(dist, torch, t_sel, dist.ReduceOp.SUM, group)
            ctx.n50_select_pick(r)
        return ctx.fetch_tables(copy=copy)
    t_sum = _tensor(ctx, torch, views, "sum")
    _all_reduce(dist, torch, t_sum, dist.ReduceOp.SUM, group)
    _cells_to_host(ctx, torch)
    n_local = views["lengths"].__cuda_array_interface__["shape"][0]
    if n_local:
        lens = _tensor(ctx, torch, views, "lengths")
    else:
        lens = torch.zeros(0, dtype=torch.int32, device="cuda")
    all_lens = gather_lengths(lens, dist, torch, group)
    ctx.finalize_n50(all_lens.data_ptr(), all_lens.numel())
    tables = ctx.fetch_tables(copy=copy)         # synchronises: all_lens may be released afterwards
    return tables


def _finalize_mode1(ctx, dist, torch, group, copy):
    """time_mode 1: extremes -> MAX; the three phases of the hour rule with their merges; tables (+ first select
    histogram) -> SUM; two more select rounds.  Seven small collectives and one table-sized one, all stream-ordered."""
    ctx.finalize_extremes()
    views, _ = ctx.device_tables()
    t_maxmin = _tensor(ctx, torch, views, "maxmin")
    _all_reduce(dist, torch, t_maxmin, dist.ReduceOp.MAX, group)
    m1 = ctx.mode1_dist_tables()
    ctx.mode1_dist_phase(0)
    _all_reduce(dist, torch, torch.as_tensor(m1["first"], device="cuda"), dist.ReduceOp.MAX, group)
    ctx.mode1_dist_phase(1)
    _all_reduce(dist, torch, torch.as_tensor(m1["ties"], device="cuda"), dist.ReduceOp.SUM, group)
    _all_reduce(dist, torch, torch.as_tensor(m1["days"], device="cuda"), dist.ReduceOp.MAX, group)
    ctx.mode1_dist_phase(2)
    _merge_compat_counts(ctx, dist, torch, group)
    ctx.finalize_tables()
    t_sum_sel = _tensor(ctx, torch, views, "sum_sel")
    t_sel = _tensor(ctx, torch, views, "sel")
    ctx.n50_select_hist(0)
    _all_reduce(dist, torch, t_sum_sel, dist.ReduceOp.SUM, group)
    ctx.n50_select_pick(0)
    for r in (1, 2):
        ctx.n50_select_hist(r)
        _all_reduce(dist, torch, t_sel, dist.ReduceOp.SUM, group)
        ctx.n50_select_pick(r)
    return ctx.fetch_tables(copy=copy)


def allreduce_min_int(value, dist, torch, group=None, device="cuda"):
    """MIN over the ranks of one host integer (RefTime of a FASTQ-only run: the earliest start time of the job)."""
    if _gloo(dist, group):
        device = "cpu"                                 # see _all_reduce
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def allreduce_sum_ints(values, dist, torch, group=None, device="cuda"):
    """SUM over the ranks of a few host integers (counts every rank keeps of its own share of a job)."""
    if _gloo(dist, group):
        device = "cpu"                                 # see _all_reduce
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(v) for v in t.tolist()]


def shard_files(files, rank, world):
    """Contiguous shard of a file list: rank r takes files [r F / N, (r + 1) F / N).  Contiguous, so that the reads of a
    channel keep their order (file order, then record order) when the ranks' outputs are put together in rank order
    (-fq).  The counterpart of the reference handing one task per channel / per file to its pool of worker processes
    (lib/seq_routines.py:909-923, 975-990)."""
    files = list(files)
    return files[rank * len(files) // world:(rank + 1) * len(files) // world]


MODE1_DAYS = 64   # PPQC_MODE1_DAYS


def mode1_hours_reference(channel, time, cfg, all_reduce_max, all_reduce_sum):
    """The arithmetic of ppqc_mode1_dist_phase 0 / 1 / 2 on host arrays (tests): ``channel`` / ``time`` are this rank's
    reads, ``all_reduce_max`` / ``all_reduce_sum`` merge an int64 numpy array across the ranks in place.  Returns this
    rank's hour bins (-1 = none) and the merged count of reads outside the day window."""
    import numpy as np
    nch, nb = int(cfg["n_channels"]), int(cfg["n_time_bins"])
    bs, ref, tz = int(cfg["bin_seconds"]), int(cfg["ref_time_s"]), int(cfg["tz_offset_s"])
    ch = np.asarray(channel, dtype=np.int64)
    t = np.asarray(time, dtype=np.int64)
    NONE = np.iinfo(np.int64).min
    ok = (ch >= 1) & (ch <= nch) & (t != NONE)
    hod = lambda x: ((x - ref + tz) // bs) % 24
    day = lambda x: (x - ref + tz) // (24 * bs) + 1
    first = np.full(nch, NONE, np.int64)                       # ~t0, unseen = ~INT64_MAX
    np.maximum.at(first, ch[ok] - 1, ~t[ok])
    all_reduce_max(first)
    ties = np.zeros(nch + 1, np.int64)
    days = np.full(2 * nch * MODE1_DAYS, NONE, np.int64)
    dmax, dmin = days[:nch * MODE1_DAYS], days[nch * MODE1_DAYS:]
    tie = ok.copy()
    tie[ok] = t[ok] == ~first[ch[ok] - 1]
    np.add.at(ties, ch[tie] - 1, 1)
    rest = ok & ~tie
    d = day(t[rest])
    inside = (d >= 0) & (d < MODE1_DAYS)
    ties[nch] += int((~inside).sum())
    k = (ch[rest][inside] - 1) * MODE1_DAYS + d[inside]
    np.maximum.at(dmax, k, t[rest][inside])
    np.maximum.at(dmin, k, ~t[rest][inside])
    all_reduce_sum(ties)
    all_reduce_max(days)
    tstar = np.full(nch, np.iinfo(np.int64).max, np.int64)
    for c in range(nch):
        if first[c] == NONE:
            continue
        t0 = int(~first[c])
        d0, m = int(day(t0)), -1
        if ties[c] >= 2 and not 0 <= d0 < MODE1_DAYS:          # reads of S at t0 that the day tables cannot hold
            ties[nch] += 1
        for dd in range(MODE1_DAYS):
            mx, mn = int(dmax[c * MODE1_DAYS + dd]), int(~dmin[c * MODE1_DAYS + dd])
            any_ = mx != NONE
            if ties[c] >= 2 and dd == d0:
                mn = t0
                mx = mx if any_ else t0
                any_ = True
            if not any_:
                continue
            if hod(mn) < m:
                tstar[c] = mn
                break
            m = max(m, int(hod(mx)))
    hours = np.full(len(t), -1, np.int32)
    h = hod(t[ok]) + np.where(t[ok] >= tstar[ch[ok] - 1], 24, 0)
    hours[ok] = np.where(h < nb, h, -1)
    return hours, int(ties[nch])


def compat_single_reference(tables, channel, mux, length, fail, hour, job_count, cfg):
    """The arithmetic of compat_single_kernel with ppqc_compat_dist_count on host arrays (tests): ``tables`` are this
    rank's tables built WITHOUT the rule from its own reads (``channel`` .. ``hour`` per read), ``job_count`` the
    all-reduced reads per channel.  A channel that holds one read in the whole job has it on one rank: that rank
    rewrites the channel's cells (chan_fail = n_mux * fail, (1, length) under every mux at the read's hour) and tops up
    the hour totals; summed over the ranks the tables are those of channel_parser on the whole job
    (lib/multi_read_fast5_reader.py:64-70).  In place."""
    nch, nmux, nb = int(cfg["n_channels"]), int(cfg["n_mux"]), int(cfg["n_time_bins"])
    for i in range(len(channel)):
        c = int(channel[i]) - 1
        if c < 0 or c >= nch or int(job_count[c]) != 1:
            continue
        L, h = int(length[i]), int(hour[i])
        tables["chan_fail"][c] = nmux * int(bool(fail[i]))
        if h < 0 or h >= nb:
            continue
        tables["cmh_reads"][c, :, h] = 1
        tables["cmh_bases"][c, :, h] = L
        extra = nmux - (1 if 1 <= int(mux[i]) <= nmux else 0)
        tables["hour_reads"][h] += extra
        tables["hour_bases"][h] += extra * L
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
