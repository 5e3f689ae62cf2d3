#!/usr/bin/env python
"""bench.py -- Gbases/s of the per-read QC statistics path (BASELINE.json metric) on N MI355X of one node.

A step = one whole pass of the hot path over one resident batch: reset, per-read segmented reduction over the CSR
batch, per-read finalisation, extremes, run-level tables, radix sort + N50 and, for N > 1, the RCCL merges
(all-reduce MAX/MIN of the extremes, all-reduce SUM of the fused table buffer, all-gather of the lengths).
Inputs are synthetic (BASELINE.md config 2 shape: 1 M reads per GPU, log-normal lengths with N50 ~ 20 kb, about
10 Gbases + 10 Gquals per GPU) and are generated in HBM before the timed region.  Weak scaling: every rank holds its
own config-2 shard.

Prints ONE JSON line on rank 0 (see the contract in DESIGN.md / the task description).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (config 2: 1e6)")
    ap.add_argument("--mu", type=float, default=8.5172)
    ap.add_argument("--sigma", type=float, default=1.1774)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--time-bins", type=int, default=48)
    ap.add_argument("--seed", type=int, default=20240601)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the short end-to-end (pinned host -> GPU) leg")
    ap.add_argument("--cpu-sample-gbases", type=float, default=0.0, help="0 = auto (2 Gbases prefix)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work timed for the cpu_baseline leg")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config4adv"])
    ap.add_argument("--stream", action="store_true",
                    help="BASELINE config 5 instead of the headline metric: the same reads streamed from pinned host "
                         "memory as 4000-read batches through the double-buffered ppqc_submit (end to end incl. H2D)")
    ap.add_argument("--batch-reads", type=int, default=4000)
    ap.add_argument("--fastq", action="store_true",
                    help="FASTQ TEXT as input instead of CSR (not the headline metric): GPU tokenizer on text resident in "
                         "HBM, the same streamed from pinned host memory file by file, and the host packer for comparison")
    ap.add_argument("--dist-path", action="store_true",
                    help="with one GPU: run the step through the multi-GPU merge path (1-rank RCCL group) to time it")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (a.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.dist_path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pypore_amd import qc, synth
    from pypore_amd.merge import finalize_distributed

    if a.workload == "config4":
        a.mu, a.reads = 10.1266, min(a.reads, 200_000)
    if a.workload == "config3":   # the 10 M-read PromethION run sharded 8 ways: 1.25 M reads per GPU
        a.mu, a.reads, a.channels, a.time_bins = 7.8240, 1_250_000, 3000, 72
    seed = a.seed + 7919 * rank
    if a.workload == "config4adv":
        a.reads = min(a.reads, 200_000)
        lens = synth.adversarial_lengths(a.reads, seed)
    else:
        lens = synth.read_lengths(a.reads, a.mu, a.sigma, seed)
    off = synth.offsets_from_lengths(lens)
    n_bytes, R = int(off[-1]), len(lens)
    ch, mux, t = synth.metadata(R, a.channels, a.time_bins, seed)
    cfg = dict(n_channels=a.channels, n_mux=4, n_time_bins=a.time_bins, bin_seconds=3600, time_mode=0, tz_offset_s=0,
               ref_time_s=1_528_000_000, fail_q=7.0, max_reads_hint=R)
    ctx = qc.QCContext(device=local, **cfg)
    if use_dist:   # one stream-ordered chain of kernels and RCCL collectives, no host synchronisation in a step
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # inputs resident in HBM before the timed region
    d_bases = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_quals = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(d_bases.data_ptr(), d_quals.data_ptr(), 0, n_bytes, seed)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ch = torch.from_numpy(ch.view(np.int32)).to(dev)
    d_mux = torch.from_numpy(mux).to(dev)
    d_t = torch.from_numpy(t).to(dev)
    ctx.sync()
    torch.cuda.synchronize()

    if a.fastq:
        return fastq_mode(a, ctx, torch, off, ch, t, d_bases, d_quals)
    if a.stream:
        return stream_mode(a, ctx, torch, off, ch, mux, t, d_bases, d_quals, n_bytes, R)

    def step():
        ctx.reset()
        ctx.submit_device(R, n_bytes, d_bases.data_ptr(), d_quals.data_ptr(), d_off.data_ptr(), d_ch.data_ptr(),
                          d_mux.data_ptr(), d_t.data_ptr())
        if use_dist:
            return finalize_distributed(ctx, dist, torch, copy=False)
        return ctx.finalize(copy=False)   # tables delivered as views of the pinned host result buffer

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tables = step()
    ctx.reset_timings()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tables = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([n_bytes], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bases = int(tot.item())
    else:
        total_bases = n_bytes
    tm = ctx.timings()

    # sanity: the merged tables must account for every read and base of the job
    assert int(tables["totals"][0]) == R * world, (tables["totals"][0], R * world)
    assert int(tables["totals"][1]) == total_bases, (tables["totals"][1], total_bases)

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = total_bases * a.steps / dt / 1e9
        launches = max(1, tm["tile_kernel_launches"])
        k_ms = tm["tile_kernel_ms_total"] / launches
        alg_bytes = tm["tile_kernel_bytes"] / launches      # 2*B + 50*R per launch (BASELINE.md accounting)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "Gbases/s per-read QC stats", "value": round(value, 2), "unit": "Gbases/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d reads/GPU, lognormal(mu=%.4f, sigma=%.4f) clipped [1, 2e6], %.2f Gbases + "
                                   "quals per GPU, %d channels, %d hourly bins, tables + radix-sort N50 included"
                                   % (a.workload, R, a.mu, a.sigma, n_bytes / 1e9, a.channels, a.time_bins),
                       "reads_per_gpu": R, "bases_per_gpu": n_bytes, "n50": int(tables["n50"]),
                       "kernel_only_gbases_per_s_per_gpu": round(n_bytes / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                       "finalize_ms": round(tm["finalize_ms_last"], 3), "sort_ms": round(tm["sort_ms_last"], 3)},
            "roofline": {"bound": "hbm", "kernel": "tile_kernel", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": None, "kernel_ms": round(k_ms, 4), "bytes_per_launch": int(alg_bytes)},
        }
        out["roofline"].update(pmc_traffic(int(alg_bytes)))
        if world == 1 and not a.no_h2d:
            out["config"]["end_to_end_incl_h2d"] = h2d_leg(a, torch, off, ch, mux, t, d_bases, d_quals, cfg)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, off, d_bases, d_quals, ctx)
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def h2d_leg(a, torch, off, ch, mux, t, d_bases, d_quals, cfg):
    """End to end including H2D on a bounded sample (BASELINE config 5 in small): the first ~2 Gbases of the same reads
    streamed from pinned host memory as 4000-read batches through the two staging slots of ppqc_submit, tables + N50
    included.  PCIe-bound; reported beside `value`, never as `value`.  `--stream` runs the whole workload this way."""
    from pypore_amd import qc
    r = int(np.searchsorted(off, np.uint64(2_000_000_000), side="right")) - 1
    r = max(1, min(r, len(off) - 1))
    nb = int(off[r])
    pb, pq = qc.PinnedBuffer(nb), qc.PinnedBuffer(nb)
    torch.from_numpy(pb.array).copy_(d_bases[:nb])
    torch.from_numpy(pq.array).copy_(d_quals[:nb])
    torch.cuda.synchronize()
    cuts = list(range(0, r, a.batch_reads)) + [r]
    batches = []
    for k in range(len(cuts) - 1):
        i, j = cuts[k], cuts[k + 1]
        s0, s1 = int(off[i]), int(off[j])
        batches.append((pb.array[s0:s1], pq.array[s0:s1], (off[i:j + 1] - off[i]).astype(np.uint64), ch[i:j], mux[i:j],
                        t[i:j]))
    c2 = dict(cfg); c2["max_reads_hint"] = r
    ctx = qc.QCContext(device=torch.cuda.current_device(), **c2)

    def step():
        ctx.reset()
        for k, b in enumerate(batches):
            ctx.submit(*b, slot=k % 2)
        return ctx.finalize(copy=False)

    step()
    ctx.sync()
    t0 = time.perf_counter()
    passes = 3
    for _ in range(passes):
        tables = step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / passes
    ok = int(tables["totals"][0]) == r and int(tables["totals"][1]) == nb
    ctx.close(); pb.close(); pq.close()
    return {"gbases_per_s": round(nb / dt / 1e9, 2), "h2d_GBps": round((2 * nb + 21 * r) / dt / 1e9, 2),
            "sample": "first %d reads (%.2f Gbases) in %d batches of %d reads from pinned host memory, two staging "
                      "slots, tables + N50 included, %d passes" % (r, nb / 1e9, len(batches), a.batch_reads, passes),
            "totals_ok": bool(ok)}


def stream_mode(a, ctx, torch, off, ch, mux, t, d_bases, d_quals, n_bytes, R):
    """BASELINE config 5: host -> GPU streaming.  The synthetic reads are first brought to pinned host memory (that
    copy is not timed), then every step submits them as `batch_reads`-read CSR batches alternating between the two
    staging slots (H2D on the slot's copy stream overlaps the kernels of the other slot) and finalises the run."""
    from pypore_amd import qc
    pb, pq = qc.PinnedBuffer(n_bytes), qc.PinnedBuffer(n_bytes)
    torch.from_numpy(pb.array).copy_(d_bases[:n_bytes])
    torch.from_numpy(pq.array).copy_(d_quals[:n_bytes])
    torch.cuda.synchronize()
    cuts = list(range(0, R, a.batch_reads)) + [R]
    batches = []
    for k in range(len(cuts) - 1):
        i, j = cuts[k], cuts[k + 1]
        s0, s1 = int(off[i]), int(off[j])
        batches.append((pb.array[s0:s1], pq.array[s0:s1], (off[i:j + 1] - off[i]).astype(np.uint64), ch[i:j], mux[i:j],
                        t[i:j]))

    def step():
        ctx.reset()
        for k, b in enumerate(batches):
            ctx.submit(*b, slot=k % 2)
        return ctx.finalize()

    for _ in range(a.warmup):
        tables = step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tables = step()
    ctx.sync()
    dt = time.perf_counter() - t0
    assert int(tables["totals"][0]) == R and int(tables["totals"][1]) == n_bytes
    ms = dt / a.steps * 1e3
    out = {"metric": "Gbases/s per-read QC stats, end to end incl. H2D (BASELINE config 5)",
           "value": round(n_bytes * a.steps / dt / 1e9, 2), "unit": "Gbases/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(ms, 3), "h2d_GBps": round((2 * n_bytes + 21 * R) * a.steps / dt / 1e9, 2),
           "config": {"workload": "%s streamed: %d reads in %d batches of %d reads from pinned host memory, two "
                                  "staging slots, tables + N50 included" % (a.workload, R, len(batches), a.batch_reads),
                      "bases": n_bytes}, "data": "synthetic", "dtype": "u8"}
    print(json.dumps(out), flush=True)
    pb.close(); pq.close()
    ctx.close()


def fastq_mode(a, ctx, torch, off, ch, t, d_bases, d_quals):
    """FASTQ text in, tables out.  The first `--reads` (capped at 200 000) synthetic reads are written as albacore-style
    4-line records, `batch_reads` per file."""
    from pypore_amd import qc, fastq_io
    R = min(len(off) - 1, 200_000)
    nb = int(off[R])
    hb = d_bases[:nb].cpu().numpy().tobytes()
    hq = d_quals[:nb].cpu().numpy().tobytes()
    files, cur = [], []
    for i in range(R):
        s0, s1 = int(off[i]), int(off[i + 1])
        cur.append(b"@read%08d runid=synthetic read=%d ch=%d start_time=%s\n%s\n+\n%s\n" % (
            i, i, int(ch[i]), time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(int(t[i]))).encode(), hb[s0:s1], hq[s0:s1]))
        if len(cur) == a.batch_reads or i == R - 1:
            files.append(b"".join(cur))
            cur = []
    del hb, hq
    total = sum(len(f) for f in files)
    pin = qc.PinnedBuffer(total)
    ext, pos = [], 0
    for f in files:
        pin.array[pos:pos + len(f)] = np.frombuffer(f, np.uint8)
        ext.append((pos, pos + len(f)))
        pos += len(f)
    d_text = torch.empty(total + 64, dtype=torch.uint8, device=d_bases.device)
    d_text[:total].copy_(torch.from_numpy(pin.array))
    torch.cuda.synchronize()
    res = {}

    def timed(name, fn):
        for _ in range(a.warmup):
            tab = fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tab = fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / a.steps
        assert int(tab["totals"][0]) == R and int(tab["totals"][1]) == nb and int(tab["totals"][7]) == 0
        res[name] = {"ms": round(dt * 1e3, 3), "Gbases_per_s": round(nb / dt / 1e9, 2), "text_GBps": round(total / dt / 1e9, 2)}
        return tab

    def resident():      # whole text already in HBM, one call
        ctx.reset()
        ctx.submit_fastq_device(d_text.data_ptr(), total)
        return ctx.finalize()

    def streamed():      # pinned host text, file by file through the two staging slots
        ctx.reset()
        for k, (s0, s1) in enumerate(ext):
            ctx.submit_fastq(pin.array[s0:s1], slot=k % 2)
        return ctx.finalize()

    def host_packer():   # ppqc_fastq_scan/pack on one host thread, then the CSR path
        ctx.reset()
        for k, (s0, s1) in enumerate(ext):
            p = fastq_io.pack_fastq(pin.array[s0:s1])
            ctx.submit(p["bases"], p["quals"], p["offsets"], p["channel"], np.ones(p["n_reads"], np.uint8),
                       p["start_time"], slot=k % 2)
        return ctx.finalize()

    ta = timed("gpu_tokenizer_text_resident_in_hbm", resident)
    tb = timed("gpu_tokenizer_streamed_from_pinned_host", streamed)
    tc = timed("host_packer_then_csr", host_packer)
    same = all(np.array_equal(ta[k], tb[k]) and np.array_equal(ta[k], tc[k]) for k in
               ("totals", "gc_hist", "q_hist", "chan_reads", "chan_bases", "chan_fail", "cmh_bases", "hour_reads"))
    print(json.dumps({"metric": "Gbases/s per-read QC stats from FASTQ text", "unit": "Gbases/s", "n_gpus": 1,
                      "steps": a.steps, "warmup": a.warmup, "data": "synthetic",
                      "config": {"workload": "%d reads, %.2f Gbases, %.2f GB of FASTQ text in %d files of %d records"
                                             % (R, nb / 1e9, total / 1e9, len(files), a.batch_reads)},
                      "results": res, "all_three_paths_identical_tables": bool(same)}), flush=True)
    pin.close()
    ctx.close()


def pmc_traffic(alg_bytes):
    """HBM bytes per tile-kernel launch from the committed rocprofv3 PMC passes of this same command (profiles/;
    2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md).  Counters cannot be collected from
    inside the timed process, so the newest profile whose algorithmic byte count equals this run's is quoted."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")), reverse=True):
        try:
            t = json.load(open(f)).get("tile_kernel_traffic_bytes_per_launch") or {}
        except (OSError, ValueError):
            continue
        if t.get("algorithmic") == alg_bytes and t.get("total"):
            return {"traffic": int(t["total"]), "traffic_source": "profiles/" + os.path.basename(f)}
    return {}


def synth_meta(a, R):
    from pypore_amd import synth
    return synth.metadata(R, a.channels, a.time_bins, a.seed)


def cpu_baseline(a, off, d_bases, d_quals, ctx):
    """The C oracle ("port" of the reference's per-read arithmetic) on the host cores, on a bounded sample of the
    same workload: a prefix of at most 2 Gbases, passed over repeatedly until about 12 s of CPU work have been
    timed.  Also re-checks the GPU records of that prefix against it."""
    from oracle import c_oracle as co
    cores = min(16, os.cpu_count() or 1)
    target = a.cpu_sample_gbases * 1e9 if a.cpu_sample_gbases > 0 else 2.0e9
    r = int(np.searchsorted(off, np.uint64(target), side="right")) - 1
    r = max(1, min(r, len(off) - 1))
    nb = int(off[r])
    hb = d_bases[:nb].cpu().numpy()
    hq = d_quals[:nb].cpu().numpy()
    sub = off[: r + 1].copy()
    want = co.per_read(hb, hq, sub, nthreads=cores)          # untimed first pass: page-in + the parity re-check
    got = ctx.read_stats(0, r)
    ok = all(np.array_equal(got[k], want[k]) for k in ("length", "gc", "n", "qsum"))
    ok = ok and [(int(h) << 32) + int(l) for h, l in zip(got["err_hi"], got["err_lo"])] == [int(e) for e in want["err"]]
    passes, dt, budget = 0, 0.0, a.cpu_seconds
    while dt < budget and passes < 1000:
        t0 = time.perf_counter()
        co.per_read(hb, hq, sub, nthreads=cores)
        dt += time.perf_counter() - t0
        passes += 1
    # the same statistics organised the way the reference organises them (pure Python, one task per channel on a
    # pool of `cores` processes), on a smaller prefix: SURVEY.md 8d's "reference-shaped" figure
    from oracle import ref_shaped
    r2 = min(r, 200_000)
    nb2 = int(off[r2])
    ch, mux, t = synth_meta(a, len(off) - 1)
    dt2, _cd = ref_shaped.run(hb[:nb2], sub[: r2 + 1], ch[:r2], mux[:r2], t[:r2], got["mean_q"][:r2], 1_528_000_000, cores)
    shaped = {"value": round(nb2 / dt2 / 1e9, 4), "unit": "Gbases/s", "cores": cores,
              "sample": "first %d reads (%.2f Gbases), oracle/ref_shaped.py: multiprocessing pool, one task per "
                        "channel, GC() as six count passes, parsing_func aggregation; HDF5 / dateutil / Biopython "
                        "record costs of the real reference excluded, %.1f s" % (r2, nb2 / 1e9, dt2)}
    return {"value": round(nb * passes / dt / 1e9, 3), "unit": "Gbases/s", "cores": cores, "kind": "port",
            "reference_shaped_python": shaped,
            "sample": "first %d reads (%.2f Gbases) of the same workload x %d passes, oracle/ppqc_oracle.c per-read "
                      "statistics, %d pthreads, %.1f s timed" % (r, nb / 1e9, passes, cores, dt),
            "gpu_matches_cpu_on_sample": bool(ok)}


if __name__ == "__main__":
    main()
