"""Regenerates the golden fixtures in this directory from the CPU oracle (oracle/pypore_oracle.py).

The reference (rsemeraro/PyPore) has no tests or fixtures and cannot be run here (Python 2; h5py / Biopython absent),
so the vectors are (a) known answers derived by hand from the cited reference lines (kat.json, written by hand, NOT
produced by this script) and (b) outputs of the oracle on small seeded inputs, pinned here so that later changes to the
oracle or to the HIP path are caught:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

from oracle import pypore_oracle as po   # noqa: E402
from util import random_batch            # noqa: E402

REF = 1_000_000_000


def toy_reads():
    """3 channels x 2 muxes: mux-0 read, fail-path read, q exactly 7.0 and just below, hours 23 -> 24 -> 25 -> 47 -> 49,
    the index-1 wrap bug (channel 2) and an empty channel (3).  (channel, mux, hours after ref, bases, quals, fail)"""
    def seq(n, gc):   # n bases with exactly gc G/C symbols
        return (b"G" * gc + b"A" * (n - gc))
    q = lambda n, v: bytes([33 + v]) * n
    return [
        (1, 1, 0.5, seq(1000, 400), q(1000, 10), 0),
        (1, 1, 23.2, seq(2000, 1000), q(2000, 7), 0),                      # mean_q == 7.0 exactly: pass
        (1, 2, 25.1, seq(3000, 1350), q(2999, 7) + q(1, 6), 0),            # just below 7.0: fail
        (1, 1, 26.5, seq(500, 150), q(500, 10), 0),
        (1, 0, 47.9, seq(700, 140), q(700, 8), 0),                         # mux 0: channel totals only
        (1, 2, 49.2, seq(50, 10), q(50, 30), 1),                           # 'fail' path: len 0, q 6.99, gc 0
        (2, 1, 23.5, seq(100, 50), q(100, 8), 0),
        (2, 1, 24.5, seq(200, 100), q(200, 8), 0),                         # index 1 is never wrap-fixed: stays hour 0
        (2, 1, 25.5, seq(300, 150), q(300, 8), 0),
    ]


def toy_batch():
    reads = toy_reads()
    bases = b"".join(r[3] for r in reads)
    quals = b"".join(r[4] for r in reads)
    off = np.concatenate(([0], np.cumsum([len(r[3]) for r in reads]))).astype(np.uint64)
    return dict(bases=np.frombuffer(bases, np.uint8), quals=np.frombuffer(quals, np.uint8), offsets=off,
                channel=np.array([r[0] for r in reads], np.uint32), mux=np.array([r[1] for r in reads], np.uint8),
                time=np.array([REF + int(r[2] * 3600) for r in reads], np.int64),
                fail_flag=np.array([r[5] for r in reads], np.uint8))


TOY_CFG = dict(n_channels=3, n_mux=4, n_time_bins=48, bin_seconds=3600, time_mode=1, tz_offset_s=0, ref_time_s=REF,
               fail_q=7.0)


def tables_to_json(t):
    out = {}
    for k, v in t.items():
        if k in ("fail", "qbin", "hour"):
            out[k] = [int(x) for x in v]
        elif isinstance(v, np.ndarray):
            out[k] = v.astype(np.int64).tolist()
        else:
            out[k] = int(v)
    return out


def main():
    b = toy_batch()
    st = po.per_read_stats(b["bases"], b["quals"], b["offsets"])
    tb = po.build_tables(st, b["channel"], b["mux"], b["time"], b["fail_flag"], TOY_CFG)
    toy = dict(per_read={k: [int(x) for x in st[k]] for k in ("length", "gc", "n", "qsum", "err")},
               mean_q=[float(x) for x in st["mean_q"]], tables=tables_to_json(tb))
    json.dump(toy, open(os.path.join(HERE, "toy_run.json"), "w"), indent=1)

    rb = random_batch(2024, 400, 300, n_channels=16, edge=False)
    st = po.per_read_stats(rb["bases"], rb["quals"], rb["offsets"])
    cfg0 = dict(TOY_CFG, n_channels=16, time_mode=0, ref_time_s=1_528_000_000)
    cfg1 = dict(cfg0, time_mode=1, tz_offset_s=7200)
    t0 = po.build_tables(st, rb["channel"], rb["mux"], rb["time"], rb["fail_flag"], cfg0)
    t1 = po.build_tables(st, rb["channel"], rb["mux"], rb["time"], rb["fail_flag"], cfg1)
    np.savez_compressed(os.path.join(HERE, "random_small.npz"), **rb,
                        exp_length=st["length"], exp_gc=st["gc"], exp_n=st["n"], exp_qsum=st["qsum"],
                        exp_err_hi=np.array([int(e) >> 32 for e in st["err"]], np.uint64),
                        exp_err_lo=np.array([int(e) & 0xFFFFFFFF for e in st["err"]], np.uint64),
                        exp_mean_q=st["mean_q"])
    json.dump(dict(cfg0=cfg0, cfg1=cfg1, tables0=tables_to_json(t0), tables1=tables_to_json(t1)),
              open(os.path.join(HERE, "random_small_tables.json"), "w"))
    print("golden fixtures written")


if __name__ == "__main__":
    main()
