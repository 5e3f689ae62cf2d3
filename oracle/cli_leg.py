"""bench.py --cli: the reference-shaped CPU figure on the same FASTQ files (TEST / BENCH INFRASTRUCTURE ONLY, see
pypore_oracle.py's header).  A program of its own, started by bench.py as a child process that never touches the GPU.

What the reference does with FASTQ input (albacore mode, lib/seq_routines.py:40-53, 896-901, 1032-1072): a pool of worker
processes, one task per file, each parsing the records (SeqIO grammar, restated in pypore_oracle.parse_fastq_records) and
computing GC() per read; then the per-channel aggregation of parsing_func over the per-read tuples.  Channel and start
time come from the FASTQ headers here (the reference takes them from sequencing_summary.txt) and the mean quality is a
constant (the reference reads it from that table too), which flatters the CPU side.  A bounded sample: the first files
of the directory up to --max-gbases."""
from __future__ import annotations

import argparse
import calendar
import json
import multiprocessing as mp
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pypore_oracle as po   # noqa: E402

_CH = re.compile(rb"ch=(\d+)")
_ST = re.compile(rb"start_time=(\d+)-(\d+)-(\d+)T(\d+):(\d+):(\d+)Z")


def _file_task(path):
    text = open(path, "rb").read()
    out = []
    for title, seq, _qual in po.parse_fastq_records(text):
        ch = int(_CH.search(title).group(1))
        t = calendar.timegm(tuple(int(x) for x in _ST.search(title).groups()) + (0, 0, 0))
        out.append(po.get_content_tuple(ch, 1, t, len(seq), 10.0, po.GC(seq)))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--max-gbases", type=float, default=0.5)
    ap.add_argument("--procs", type=int, default=0)
    a = ap.parse_args(argv)
    files = sorted(os.path.join(a.directory, f) for f in os.listdir(a.directory) if f.endswith(".fastq"))
    sample, size = [], 0
    for f in files:      # text bytes are about twice the bases
        sample.append(f)
        size += os.path.getsize(f)
        if size >= 2.0 * a.max_gbases * 1e9:
            break
    share = os.environ.get("OMP_NUM_THREADS", "")
    procs = a.procs or (int(share) if share.isdigit() and int(share) > 0 else len(os.sched_getaffinity(0)))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:      # this process has no GPU state to fork
        per_file = pool.map(_file_task, sample, chunksize=1)
    tuples = [t for part in per_file for t in part]
    ref = min(float(t[2]) for t in tuples)
    cd = po.build_channel_dict(tuples, ref, n_channels=512, tz_offset_s=0)
    st = po.FastStats_data(cd)
    dt = time.perf_counter() - t0
    bases = sum(int(t[3]) for t in tuples)
    print(json.dumps({"seconds": round(dt, 3), "gbases_per_s": round(bases / dt / 1e9, 3), "reads": len(tuples), "bases": bases,
                      "files": len(sample), "procs": procs, "table": st["table"][:4],
                      "what": "oracle/cli_leg.py: pool of %d processes, one task per FASTQ file (record parsing + GC() per "
                              "read), then parsing_func per channel and FastStats; mean quality taken as given" % procs}))


if __name__ == "__main__":
    main()
