#!/usr/bin/env python
"""bench.py -- Gbases/s of the per-read QC statistics path (BASELINE.json metric) on N MI355X of one node.

A step = one whole pass of the hot path over one resident batch: reset, per-read segmented reduction over the CSR
batch, per-read finalisation, extremes, run-level tables, the exact N50 (radix select) and, for N > 1, the RCCL merges
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="keep timing further blocks of exactly --steps steps until this much time has been timed "
                         "(every block bracketed the same way; the figures are means over the blocks)")
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (config 2: 1e6)")
    ap.add_argument("--mu", type=float, default=8.5172)
    ap.add_argument("--sigma", type=float, default=1.1774)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--time-bins", type=int, default=48)
    ap.add_argument("--seed", type=int, default=20240601)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the short end-to-end (pinned host -> GPU) leg")
    ap.add_argument("--no-config3", action="store_true", help="N > 1: skip the config-3 sharded leg")
    ap.add_argument("--merge-behind", action="store_true",
                    help="A/B: the merge collectives behind the per-read kernel (finalize_distributed(early=False)) "
                         "instead of beside it on a side stream")
    ap.add_argument("--config3-reads", type=int, default=0, help="reads of the config-3 job (0: the 10 M of BASELINE config 3)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="NOT a measurement: the N ranks share GPU 0 and talk over gloo, so that the whole N > 1 code path "
                         "of this script (launch, sharding, merges, both workloads, reporting) can be run on a one-GPU box")
    ap.add_argument("--config3-leg", action="store_true",
                    help="with --dist-path on one GPU: run the config-3 sharded leg too (the whole 10 M-read job on one "
                         "rank, about 100 GB of HBM), to rehearse the N > 1 code path")
    ap.add_argument("--cpu-sample-gbases", type=float, default=0.0, help="0 = auto (2 Gbases prefix)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work timed for the cpu_baseline leg")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config4adv"])
    ap.add_argument("--stream", action="store_true",
                    help="BASELINE config 5 instead of the headline metric: the same reads streamed from pinned host "
                         "memory as 4000-read batches through the double-buffered ppqc_submit (end to end incl. H2D)")
    ap.add_argument("--batch-reads", type=int, default=4000)
    ap.add_argument("--fastq-reads", type=int, default=200_000, help="with --fastq: records of the text (the first of the workload's reads)")
    ap.add_argument("--fastq-resident-only", action="store_true", help="with --fastq: only the leg with the text resident in HBM (A/B runs)")
    ap.add_argument("--fastq", action="store_true",
                    help="FASTQ TEXT as input instead of CSR (not the headline metric): GPU tokenizer on text resident in "
                         "HBM, the same streamed from pinned host memory file by file, and the host packer for comparison")
    ap.add_argument("--cli", action="store_true",
                    help="wall clock of `pypore seqstats` (albacore FASTQ mode) over a synthetic FASTQ directory on tmpfs, "
                         "beside the reference-shaped Python on the same files (not the headline metric)")
    ap.add_argument("--cli-gbases", type=float, default=2.0)
    ap.add_argument("--cli-threads", type=int, default=4, help="--cli: the command's -n (loader threads)")
    ap.add_argument("--dist-path", action="store_true",
                    help="with one GPU: run the step through the multi-GPU merge path (1-rank RCCL group) to time it")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: N ranks over gloo go through the launch, the rendezvous, the barriers and the "
                         "max-over-ranks / sum-over-ranks reporting of this script (the CPU test of the launcher path)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="A/B runs: ppqc_set_tuning on every context of this run (e.g. tile_all=0, fastq_sums=2); results "
                         "do not depend on it")
    ap.add_argument("--master-port", type=int, default=0, help="launcher: rendezvous port (0 = pick a free one)")
    return ap.parse_args(argv)


def launch_ranks(a):
    """`bench.py --gpus N` started plainly (no WORLD_SIZE): start the N rank processes ourselves, as one torchrun child
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`, what the driver uses).  Decided before this
    process has imported torch or touched HIP; the child is a child process, never an exec of this one, and its exit
    code is ours -- a rank that fails makes the whole command fail."""
    import socket
    import subprocess
    port = a.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    argv = [x for x in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d)\n" % (a.gpus, rc))
    return rc if rc >= 0 else 1


def launcher_selftest(a, world, rank):
    """The launcher path without a GPU: rendezvous, barrier-bracketed timing, MAX over ranks of the time and SUM over
    ranks of the work -- everything main() does around the steps, with nothing computed.  Prints the JSON line with the
    number of ranks the process group really has."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pass
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    units = torch.tensor([1000 + rank], dtype=torch.int64)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    ranks = dist.get_world_size()
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest (no GPU work)", "value": None, "n_gpus": world, "rccl_ranks": ranks,
                          "backend": "gloo", "steps": a.steps, "warmup": a.warmup, "selftest": True,
                          "units_all_ranks": int(units.item()), "max_seconds": float(dt.item())}), flush=True)
    dist.destroy_process_group()
    return 0


def start_cpu_leg(a):
    """The cpu_baseline leg as a child process, started BEFORE this process touches the GPU (oracle/cpu_leg.py)."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "ppqc_cpu_leg_%d.npz" % os.getpid())
    cmd = [sys.executable, "-m", "oracle.cpu_leg", "--workload", a.workload, "--reads", str(a.reads), "--mu", str(a.mu),
           "--sigma", str(a.sigma), "--seed", str(a.seed), "--channels", str(a.channels), "--time-bins", str(a.time_bins),
           "--sample-gbases", str(a.cpu_sample_gbases if a.cpu_sample_gbases > 0 else 2.0), "--seconds",
           str(a.cpu_seconds), "--out", out]
    rc = subprocess.call(cmd, cwd=ROOT)
    if rc != 0:
        raise SystemExit("bench.py: the cpu_baseline child failed (exit code %d)" % rc)
    return out


def workload_shape(a):
    """Applies --workload to the arguments (per-GPU weak-scaling shapes; config 3's real 8-way sharding is the
    config3_sharded leg of N > 1 runs)."""
    if a.workload == "config4":
        a.mu, a.reads = 10.1266, min(a.reads, 200_000)
    if a.workload == "config3":   # the 10 M-read PromethION run sharded 8 ways: 1.25 M reads per GPU
        a.mu, a.reads, a.channels, a.time_bins = 7.8240, 1_250_000, 3000, 72
    if a.workload == "config4adv":
        a.reads = min(a.reads, 200_000)


def workload_text(a, R, n_bytes):
    if a.workload == "config4adv":
        shape = "99.95 % reads of 200 b + 100 reads of 2 Mb scattered through the batch"
    else:
        shape = "lognormal(mu=%.4f, sigma=%.4f) clipped [1, 2e6]" % (a.mu, a.sigma)
    return "%s: %d reads/GPU, %s, %.2f Gbases + quals per GPU, %d channels, %d hourly bins, tables + exact N50 " \
           "included" % (a.workload, R, shape, n_bytes / 1e9, a.channels, a.time_bins)


def timed_blocks(step, fence, steps, min_seconds, reduce_max):
    """Blocks of EXACTLY `steps` steps, each bracketed by fence() (barrier + synchronise) on both sides and reduced by
    MAX over the ranks; further blocks until `min_seconds` have been timed, decided on the reduced time so that every
    rank runs the same number.  Returns (mean seconds per block, blocks, last tables).

    The objects alive when the timing starts (torch, numpy, the workload) are moved out of the garbage collector's reach
    first (gc.freeze): the collector stays on, but a collection that the step's own short-lived objects trigger then
    looks at those only.  Without it the multi-GPU step -- six collectives, a few dozen tensors and views per step --
    triggered a collection over everything torch had imported every fifth step, 1.3 ms of host time in front of the next
    launch (measured: steps of 3.27, 3.27, 3.27, 4.55, 3.27 ms in every block); timeit switches the collector off for the
    same reason."""
    import gc
    gc.collect()
    gc.freeze()
    total, blocks, tables = 0.0, 0, None
    while True:
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            tables = step()
        fence()
        total += reduce_max(time.perf_counter() - t0)
        blocks += 1
        if total >= min_seconds or blocks >= 10_000:
            return total / blocks, blocks, tables


def main():
    a = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if a.cli and world_env is None:
        return cli_mode(a)              # the command under test starts its own ranks (`pypore seqstats --gpus N`)
    if world_env is None and a.gpus > 1:
        return launch_ranks(a)          # nothing below has run: no torch import, no HIP call in this process
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: refusing to run (launch as `python bench.py --gpus N` "
                         "or through torchrun with --nproc-per-node N)\n" % (a.gpus, world))
        return 2
    if a.launcher_selftest:
        return launcher_selftest(a, world, rank)
    if a.cli:
        return cli_mode(a)
    workload_shape(a)
    cpu_leg_file = None
    if world == 1 and not a.no_cpu_baseline and not (a.fastq or a.stream):
        cpu_leg_file = start_cpu_leg(a)       # before torch / HIP are initialised in this process

    import torch
    import torch.distributed as dist

    n_dev = torch.cuda.device_count()
    if a.rehearse_gloo:
        local = 0                      # every rank on GPU 0
    if n_dev < 1 or (not a.rehearse_gloo and (n_dev < world or local >= n_dev)):
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s): refusing to run\n" % (a.gpus, n_dev))
        return 3
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.dist_path
    rccl_ranks = 1
    cdev = torch.device("cpu") if a.rehearse_gloo else dev    # where this script's own small collectives live
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.rehearse_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        one = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)      # the ranks RCCL really connects, not the ones asked for
        rccl_ranks = int(one.item())
        if rccl_ranks != world or dist.get_world_size() != world:
            sys.stderr.write("bench.py: RCCL sees %d ranks, --gpus %d\n" % (rccl_ranks, world))
            return 4

    from pypore_amd import qc, synth
    from pypore_amd.merge import finalize_distributed
    qc.QCContext.DEFAULT_TUNING = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tune}

    seed = a.seed + 7919 * rank
    if a.workload == "config4adv":
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
        from pypore_amd.merge import _beside_stream
        _beside_stream(torch)   # the side stream's probe (a few sleeps of a millisecond) belongs to the set-up, not to a step

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
            return finalize_distributed(ctx, dist, torch, copy=False, early=False if a.merge_behind else None)
        return ctx.finalize(copy=False)   # tables delivered as views of the pinned host result buffer

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if world == 1:
            return x
        tmax = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    for _ in range(a.warmup):
        tables = step()
    ctx.reset_timings()
    dt, blocks, tables = timed_blocks(step, fence, a.steps, a.min_seconds, reduce_max)
    if world > 1:
        tot = torch.tensor([n_bytes], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bases = int(tot.item())
    else:
        total_bases = n_bytes
    tm = ctx.timings()

    # sanity: the merged tables must account for every read and base of the job
    assert int(tables["totals"][0]) == R * world, (tables["totals"][0], R * world)
    assert int(tables["totals"][1]) == total_bases, (tables["totals"][1], total_bases)
    n50_job = int(tables["n50"])

    out = None
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = total_bases * a.steps / dt / 1e9
        launches = max(1, tm["tile_kernel_launches"])
        k_ms = tm["tile_kernel_ms_total"] / launches
        alg_bytes = tm["tile_kernel_bytes"] / launches      # 2*B + 50*R per launch (BASELINE.md accounting)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "Gbases/s per-read QC stats" if not a.rehearse_gloo else
                      "REHEARSAL, not a measurement: %d ranks sharing ONE GPU over gloo" % world,
            "value": round(value, 2), "unit": "Gbases/s", "n_gpus": world,
            "rccl_ranks": rccl_ranks if use_dist else 0, "backend": "gloo (rehearsal)" if a.rehearse_gloo else "nccl",
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "timed_blocks": blocks, "timed_seconds": round(dt * blocks, 3),
            "config": {"workload": workload_text(a, R, n_bytes),
                       "reads_per_gpu": R, "bases_per_gpu": n_bytes, "n50": n50_job,
                       "kernel_only_gbases_per_s_per_gpu": round(n_bytes / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                       "finalize_ms": round(tm["finalize_ms_last"], 3), "sort_ms": round(tm["sort_ms_last"], 3)},
            "roofline": {"bound": "hbm", "kernel": "tile_kernel", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": None, "kernel_ms": round(k_ms, 4), "bytes_per_launch": int(alg_bytes),
                         "kernel_launches_timed": int(launches)},
        }
        out["roofline"].update(pmc_traffic(int(alg_bytes)))
        if world == 1 and not a.no_h2d:
            out["config"]["end_to_end_incl_h2d"] = h2d_leg(a, torch, off, ch, mux, t, d_bases, d_quals, cfg)
        if cpu_leg_file:
            out["cpu_baseline"] = cpu_baseline(cpu_leg_file, ctx)
    ctx.close()
    del d_bases, d_quals, d_off, d_ch, d_mux, d_t
    if (world > 1 and not a.no_config3) or (use_dist and a.config3_leg):
        try:
            c3 = config3_sharded(a, torch, dist, qc, synth, finalize_distributed, dev, local, rank, world)
        except Exception as e:      # the headline line must survive a failure of the second workload
            c3 = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            out["config3_sharded"] = c3
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return 0


def config3_sharded(a, torch, dist, qc, synth, finalize_distributed, dev, local, rank, world):
    """BASELINE config 3 as BASELINE.json words it: ONE 10 M-read PromethION-scale job (3000 channels, 72 hourly bins,
    lognormal N50 ~ 10 kb, about 50 Gbases) sharded CONTIGUOUSLY over the ranks -- rank r holds reads
    [r R / N, (r + 1) R / N) of the job and exactly the job's bytes for them -- merged per step by the RCCL path
    (finalize_distributed: one fused 13.9 MB all-reduce SUM of the tables behind the per-read kernel, the small MAX and
    the three N50-select rounds beside it).  Strong scaling: the job is the same for every N, and so are its tables
    (n50, totals): compare them across the SCALE runs."""
    c3 = synth.CONFIGS["config3"]
    Rj = a.config3_reads or c3["n_reads"]
    lens = synth.read_lengths(Rj, c3["mu"], c3["sigma"], a.seed)
    off = synth.offsets_from_lengths(lens)
    ch, mux, t = synth.metadata(Rj, c3["n_channels"], c3["hours"], a.seed)
    r0, r1 = rank * Rj // world, (rank + 1) * Rj // world
    b0, b1 = int(off[r0]), int(off[r1])
    R, n_bytes, job_bases = r1 - r0, b1 - b0, int(off[-1])
    cfg = dict(n_channels=c3["n_channels"], n_mux=4, n_time_bins=c3["n_time_bins"], bin_seconds=3600, time_mode=0,
               tz_offset_s=0, ref_time_s=1_528_000_000, fail_q=7.0, max_reads_hint=R)
    ctx = qc.QCContext(device=local, **cfg)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    from pypore_amd.merge import _beside_stream
    _beside_stream(torch)       # (set-up, as in the headline leg)
    # byte i of the job's streams depends on (seed, i) alone, so the shard's bytes are generated in place from b0 on
    d_bases = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_quals = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(d_bases.data_ptr(), d_quals.data_ptr(), b0, n_bytes, a.seed)
    d_off = torch.from_numpy((off[r0:r1 + 1] - np.uint64(b0)).view(np.int64)).to(dev)
    d_ch = torch.from_numpy(ch[r0:r1].view(np.int32)).to(dev)
    d_mux = torch.from_numpy(mux[r0:r1]).to(dev)
    d_t = torch.from_numpy(t[r0:r1]).to(dev)
    ctx.sync()
    torch.cuda.synchronize()

    def step():
        ctx.reset()
        ctx.submit_device(R, n_bytes, d_bases.data_ptr(), d_quals.data_ptr(), d_off.data_ptr(), d_ch.data_ptr(),
                          d_mux.data_ptr(), d_t.data_ptr())
        return finalize_distributed(ctx, dist, torch, copy=False, early=False if a.merge_behind else None)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        tmax = torch.tensor([x], dtype=torch.float64, device=torch.device("cpu") if a.rehearse_gloo else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    for _ in range(a.warmup):
        step()
    ctx.reset_timings()
    dt, blocks, tables = timed_blocks(step, fence, a.steps, a.min_seconds, reduce_max)
    tm = ctx.timings()
    # every read and base of the job, whatever the sharding
    assert int(tables["totals"][0]) == Rj and int(tables["totals"][1]) == job_bases, tables["totals"][:8]
    views, d = ctx.device_tables()
    launches = max(1, tm["tile_kernel_launches"])
    k_ms = tm["tile_kernel_ms_total"] / launches
    res = {"workload": "config3: ONE job of %d reads, %.2f Gbases, lognormal(mu=%.4f, sigma=%.4f), %d channels x 4 mux x "
                       "%d hourly bins, sharded contiguously over %d ranks (%d reads on rank 0)"
                       % (Rj, job_bases / 1e9, c3["mu"], c3["sigma"], c3["n_channels"], c3["n_time_bins"], world, R),
           "value": round(job_bases * a.steps / dt / 1e9, 2), "unit": "Gbases/s", "scaling": "strong", "n_gpus": world,
           "ms_per_step": round(dt / a.steps * 1e3, 3), "steps": a.steps, "timed_blocks": blocks,
           "table_allreduce_bytes": int(d.n_sum) * 8, "n50": int(tables["n50"]), "reads": Rj, "bases": job_bases,
           "n_fail": int(tables["totals"][2]), "tile_kernel_ms_rank0": round(k_ms, 4),
           "tile_kernel_GBps_rank0": round(tm["tile_kernel_bytes"] / launches / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None}
    ctx.close()
    return res


def cli_mode(a):
    """Wall clock of the drop-in itself: `pypore seqstats -i DIR -l bench -f fastq` (a child process: interpreter start,
    imports, loader thread reading every file into pinned slots, GPU tokenizer + kernels, tables, both HTML reports)
    over a synthetic directory of albacore-style FASTQ files on tmpfs, beside the reference-shaped Python
    (oracle/cli_leg.py) on a bounded sample of the same files.  Not the headline metric: the CLI is bound by reading
    the files and by PCIe, two orders of magnitude below the kernel's roofline."""
    import shutil
    import subprocess
    import tempfile
    from pypore_amd import synth
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="ppqc_cli_", dir=base)
    try:
        rng = np.random.default_rng(a.seed)
        lens = synth.read_lengths(a.reads, a.mu, a.sigma, a.seed)
        keep = int(np.searchsorted(np.cumsum(lens), a.cli_gbases * 1e9)) + 1
        lens = lens[:keep]
        R, nb = len(lens), int(lens.sum())
        ch, _mux, t = synth.metadata(R, a.channels, a.time_bins, a.seed)
        os.makedirs(os.path.join(work, "in"))
        alphabet = np.frombuffer(b"ACGT", np.uint8)
        n_files, text_bytes = 0, 0
        for f0 in range(0, R, a.batch_reads):
            f1 = min(R, f0 + a.batch_reads)
            n = int(lens[f0:f1].sum())
            seq = alphabet[rng.integers(0, 4, n, dtype=np.uint8)].tobytes()
            qual = rng.integers(35, 75, n, dtype=np.uint8).tobytes()
            parts, o = [], 0
            for i in range(f0, f1):
                L = int(lens[i])
                parts.append(b"@read%08d runid=synthetic read=%d ch=%d start_time=%s\n%s\n+\n%s\n" % (
                    i, i, int(ch[i]), time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(int(t[i]))).encode(),
                    seq[o:o + L], qual[o:o + L]))
                o += L
            blob = b"".join(parts)
            with open(os.path.join(work, "in", "f%05d.fastq" % n_files), "wb") as fh:
                fh.write(blob)
            n_files += 1
            text_bytes += len(blob)
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        cmd = [sys.executable, "-m", "pypore_amd.pypore", "seqstats", "-i", os.path.join(work, "in"), "-l", "bench", "-f",
               "fastq", "-o", "out", "-v", "0", "-n", str(a.cli_threads)] + (["--gpus", str(a.gpus)] if a.gpus > 1 else [])
        walls = []
        for _ in range(max(1, a.steps if a.steps < 4 else 3)):
            t0 = time.perf_counter()
            rc = subprocess.call(cmd, cwd=work, env=env, stdout=subprocess.DEVNULL)
            walls.append(time.perf_counter() - t0)
            if rc != 0:
                raise SystemExit("bench.py --cli: pypore seqstats failed (exit code %d)" % rc)
        wall = min(walls)
        ref = subprocess.run([sys.executable, "-m", "oracle.cli_leg", os.path.join(work, "in"), "--max-gbases", "0.5"],
                             cwd=ROOT, capture_output=True, text=True)
        ref_res = json.loads(ref.stdout.strip().splitlines()[-1]) if ref.returncode == 0 and ref.stdout.strip() else \
            {"error": ref.stderr[-500:]}
        out = {"metric": "Gbases/s of `pypore seqstats -f fastq`, wall clock of the whole command (not the headline metric)",
               "value": round(nb / wall / 1e9, 3), "unit": "Gbases/s", "n_gpus": a.gpus, "wall_seconds": round(wall, 3),
               "wall_seconds_all_runs": [round(w, 3) for w in walls], "file_GBps": round(text_bytes / wall / 1e9, 3),
               "data": "synthetic", "dtype": "u8",
               "config": {"workload": "%d reads, %.2f Gbases, %.2f GB of FASTQ text in %d files of %d records under %s; "
                                      "-n %d; summary + pore-activity HTML written" % (R, nb / 1e9, text_bytes / 1e9,
                                                                                        n_files, a.batch_reads, base,
                                                                                        a.cli_threads)},
               "reference_shaped_cpu": ref_res}
        if os.environ.get("PPQC_CLI_BREAKDOWN"):     # where the wall clock goes (tools/cli_breakdown.py), same files
            subprocess.call([sys.executable, os.path.join(ROOT, "tools", "cli_breakdown.py"), os.path.join(work, "in"),
                             str(a.cli_threads)],
                            cwd=work, env=env)
        print(json.dumps(out), flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return 0


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
            "totals_ok": bool(ok),
            "what": "a bounded sample of BASELINE config 5 (end to end incl. H2D): the fixed tail of a step weighs on so "
                    "short a run; the config-5 figure proper is `bench.py --stream` over the whole workload (DESIGN.md "
                    "section 6 quotes both)"}


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
    R = min(len(off) - 1, a.fastq_reads)
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
        if not os.environ.get("PPQC_BENCH_NOCHECK"):   # knock-out builds of the kernels (tools/ab_fastq.sh) give wrong sums
            assert int(tab["totals"][0]) == R and int(tab["totals"][1]) == nb and int(tab["totals"][7]) == 0
        res[name] = {"ms": round(dt * 1e3, 3), "Gbases_per_s": round(nb / dt / 1e9, 2), "text_GBps": round(total / dt / 1e9, 2)}
        return {k: np.array(v) for k, v in tab.items()}   # the next path overwrites the pinned buffer the views show

    def resident():      # whole text already in HBM, one call
        ctx.reset()
        ctx.submit_fastq_device(d_text.data_ptr(), total)
        return ctx.finalize(copy=False)   # like the headline step: the tables as views of the pinned result buffer

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
    tb = ta if a.fastq_resident_only else timed("gpu_tokenizer_streamed_from_pinned_host", streamed)
    tc = ta if a.fastq_resident_only else timed("host_packer_then_csr", host_packer)
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
    inside the timed process, so a committed profile is quoted -- but only one taken from a library built from the very
    sources this one was built from (csrc_sha16) on the same algorithmic byte count; otherwise `traffic` stays null."""
    import glob
    from pypore_amd import build
    sha = build.source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")), reverse=True):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        t = j.get("tile_kernel_traffic_bytes_per_launch") or {}
        if j.get("csrc_sha16") == sha and t.get("algorithmic") == alg_bytes and t.get("total"):
            return {"traffic": int(t["total"]), "traffic_source": "profiles/" + os.path.basename(f), "csrc_sha16": sha}
    return {"traffic": None, "traffic_note": "no committed PMC profile of these sources (csrc_sha16 %s)" % sha}


def cpu_baseline(leg_file, ctx):
    """The figures of the cpu_baseline child (oracle/cpu_leg.py, run before the GPU was touched) plus the re-check of the
    GPU's per-read records of the same prefix against the oracle's arrays the child left."""
    z = np.load(leg_file)
    res = json.loads(bytes(z["result"]).decode())
    r = int(res["prefix_reads"])
    got = ctx.read_stats(0, r)
    ok = all(np.array_equal(got[k], z[k]) for k in ("length", "gc", "n", "qsum"))
    want_err = [(int(h) << 64) | int(l) for h, l in zip(z["err_hi"], z["err_lo"])]
    ok = ok and [(int(h) << 32) + int(l) for h, l in zip(got["err_hi"], got["err_lo"])] == want_err
    res["gpu_matches_cpu_on_sample"] = bool(ok)
    try:
        os.remove(leg_file)
    except OSError:
        pass
    return res


if __name__ == "__main__":
    sys.exit(main())
