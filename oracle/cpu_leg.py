"""bench.py's cpu_baseline leg, as a program of its own (TEST / BENCH INFRASTRUCTURE ONLY, see ppqc_oracle.c's header).

bench.py starts it as a child process BEFORE it touches the GPU (no fork of a GPU-initialised process, no exec from
one), waits for it, and later compares the GPU's per-read records of the same prefix with the arrays left here.  The
prefix of the synthetic workload is regenerated on the host (oracle/ppqc_oracle.c:orc_synth_fill, the twin of the HBM
generator), so this process never initialises HIP or imports torch.

Two figures on the box's host cores: the C port of the per-read arithmetic (`kind: "port"`), and the same statistics
organised the way the reference organises them (oracle/ref_shaped.py: a pool of processes, one task per channel)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cores_of_this_box():
    """(available, used, why).  On the GPU boxes os.cpu_count() is the whole machine, many times this job's share; the
    share is what the harness exports as OMP_NUM_THREADS (16 per GPU).  Nothing is capped silently: all three are reported."""
    avail = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = avail
    share = os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and int(share) > 0:
        return avail, min(int(share), aff), "OMP_NUM_THREADS=%s (this job's CPU share on the box)" % share
    return avail, aff, "sched_getaffinity"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--reads", type=int, required=True)
    ap.add_argument("--mu", type=float, required=True)
    ap.add_argument("--sigma", type=float, required=True)
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--time-bins", type=int, default=48)
    ap.add_argument("--sample-gbases", type=float, default=2.0)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--cores", type=int, default=0, help="0 = this job's share of the box")
    ap.add_argument("--out", required=True, help="npz with the oracle's per-read arrays of the prefix + the JSON result")
    a = ap.parse_args(argv)

    from pypore_amd import synth          # numpy only: lengths, offsets, metadata
    from oracle import c_oracle as co, ref_shaped

    avail, used, why = cores_of_this_box()
    if a.cores > 0:
        used, why = a.cores, "--cores"
    lens = synth.adversarial_lengths(a.reads, a.seed) if a.workload == "config4adv" else \
        synth.read_lengths(a.reads, a.mu, a.sigma, a.seed)
    off = synth.offsets_from_lengths(lens)
    r = int(np.searchsorted(off, np.uint64(a.sample_gbases * 1e9), side="right")) - 1
    r = max(1, min(r, len(off) - 1))
    nb = int(off[r])
    sub = off[: r + 1].copy()
    hb, hq = co.synth_fill(0, nb, a.seed, used)
    want = co.per_read(hb, hq, sub, nthreads=used)      # untimed first pass (page-in); kept for the GPU re-check
    passes, dt = 0, 0.0
    while dt < a.seconds and passes < 1000:
        t0 = time.perf_counter()
        co.per_read(hb, hq, sub, nthreads=used)
        dt += time.perf_counter() - t0
        passes += 1
    # reference-shaped Python on a smaller prefix
    r2 = min(r, 200_000)
    nb2 = int(off[r2])
    ch, mux, t = synth.metadata(a.reads, a.channels, a.time_bins, a.seed)
    dt2, _cd = ref_shaped.run(hb[:nb2], sub[: r2 + 1], ch[:r2], mux[:r2], t[:r2], want["mean_q"][:r2], 1_528_000_000, used)
    res = {"value": round(nb * passes / dt / 1e9, 3), "unit": "Gbases/s", "cores": used, "cores_used": used,
           "cores_available": avail, "cores_rule": why, "kind": "port",
           "reference_shaped_python": {
               "value": round(nb2 / dt2 / 1e9, 4), "unit": "Gbases/s", "cores": used,
               "sample": "first %d reads (%.2f Gbases), oracle/ref_shaped.py: multiprocessing pool, one task per channel, "
                         "GC() as six count passes, parsing_func aggregation; HDF5 / dateutil / Biopython record costs "
                         "of the real reference excluded, %.1f s" % (r2, nb2 / 1e9, dt2)},
           "sample": "first %d reads (%.2f Gbases) of the same workload, regenerated on the host, x %d passes, "
                     "oracle/ppqc_oracle.c per-read statistics, %d pthreads, %.1f s timed, in a child process started "
                     "before the GPU was touched" % (r, nb / 1e9, passes, used, dt),
           "prefix_reads": r, "prefix_bases": nb}
    np.savez(a.out, result=np.frombuffer(json.dumps(res).encode(), np.uint8), length=want["length"], gc=want["gc"],
             n=want["n"], qsum=want["qsum"], err_hi=want["err_hi"], err_lo=want["err_lo"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
