#!/usr/bin/env python
"""Summarise a gpurun_out/prof* directory (rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes) into profiles/.

usage: python tools/summarize_prof.py gpurun_out/prof2 profiles/r01b_bench_config2
Writes <prefix>_kernel_stats.csv (copy of the rocprofv3 stats) and <prefix>_pmc_hbm.json (per-kernel mean KB of the
two PMC passes and the tile kernel's HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of a 16 B/lane streaming read)."""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict


def pmc_means(d):
    out = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            out[k][0] += 1
            out[k][1] += float(row["Counter_Value"])
    return {k: {"dispatches": n, "mean_KB": s / n} for k, (n, s) in out.items()}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    alg = int(sys.argv[3]) if len(sys.argv) > 3 else None
    ks = glob.glob(src + "/kt/**/*kernel_stats.csv", recursive=True)
    if ks:
        shutil.copy(ks[0], prefix + "_kernel_stats.csv")
    fetch, write = pmc_means(src + "/fetch"), pmc_means(src + "/write")
    res = {"FETCH_SIZE": fetch, "WRITE_SIZE": write}
    tk = [k for k in fetch if "tile_kernel" in k]
    if tk:
        fr, wr = fetch[tk[0]]["mean_KB"] * 1024, write[tk[0]]["mean_KB"] * 1024
        res["tile_kernel_traffic_bytes_per_launch"] = {"fetch_raw": fr, "fetch_corrected_x2_gfx950": 2 * fr, "write": wr,
                                                       "total": 2 * fr + wr, "algorithmic": alg}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pypore_amd import build
    res["csrc_sha16"] = build.source_hash()   # the sources the profiled library was built from (run from the same tree)
    json.dump(res, open(prefix + "_pmc_hbm.json", "w"), indent=1)
    print(json.dumps(res.get("tile_kernel_traffic_bytes_per_launch"), indent=1))
    if ks:
        print(open(ks[0]).read()[:3000])


if __name__ == "__main__":
    main()
