#!/usr/bin/env python
"""Summarise a gpurun_out/prof* directory (tools/profile_round.sh: rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes
and the SQ pass) into profiles/.

usage: python tools/summarize_prof.py gpurun_out/prof2 profiles/r02_bench_config2 [algorithmic bytes per launch]
Writes <prefix>_kernel_stats.csv (copy of the rocprofv3 stats), <prefix>_pmc_hbm.json (per-kernel mean KB of the two PMC
passes and the per-read kernels' HBM traffic per step = 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of a 16 B/lane streaming read), <prefix>_pmc_sq.json
(per-launch means of the SQ counters for the per-read kernels) and <prefix>_bench_line.json (the JSON line bench.py
printed under the kernel trace, i.e. the hipEvent figures of the very run the stats come from).  The per-read kernels are
tile_kernel and, where a batch has dense tiles, dense_kernel: one launch of each per step, timed together by the
library's events, so their counters are added per step."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

PER_READ = ("tile_kernel", "dense_kernel")


def pmc_rows(d):
    out = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            c = out[k][row["Counter_Name"]]
            c[0] += 1
            c[1] += float(row["Counter_Value"])
    return out


def pmc_means(d, counter):
    return {k: {"dispatches": v[counter][0], "mean_KB": v[counter][1] / v[counter][0]}
            for k, v in pmc_rows(d).items() if counter in v}


def per_step(means):
    """Sum over the per-read kernels of (mean x dispatches), per tile_kernel dispatch (= per step)."""
    # one launch per step of tile_kernel (with dense_kernel behind it) or, for a batch that is dense on average, of
    # dense_kernel alone
    steps = max([v["dispatches"] for k, v in means.items() if any(p in k for p in PER_READ)] or [0])
    if not steps:
        return None
    return sum(v["mean_KB"] * v["dispatches"] for k, v in means.items() if any(p in k for p in PER_READ)) * 1024 / steps


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    alg = int(sys.argv[3]) if len(sys.argv) > 3 else None
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pypore_amd import build
    sha = build.source_hash()   # the sources the profiled library was built from (run from the same tree)
    ks = sorted(glob.glob(src + "/kt/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime, reverse=True)
    if ks:
        shutil.copy(ks[0], prefix + "_kernel_stats.csv")
    line = None
    if os.path.exists(src + ".kt.log"):
        for l in open(src + ".kt.log"):
            if l.startswith("{"):
                line = json.loads(l)
        if line is not None:
            if alg is None and "roofline" in line:
                alg = line["roofline"]["bytes_per_launch"]
            json.dump(line, open(prefix + "_bench_line.json", "w"), indent=1)
    fetch, write = pmc_means(src + "/fetch", "FETCH_SIZE"), pmc_means(src + "/write", "WRITE_SIZE")
    res = {"FETCH_SIZE": fetch, "WRITE_SIZE": write}
    fr, wr = per_step(fetch), per_step(write)
    if fr is not None and wr is not None:
        res["tile_kernel_traffic_bytes_per_launch"] = {"fetch_raw": fr, "fetch_corrected_x2_gfx950": 2 * fr, "write": wr,
                                                       "total": 2 * fr + wr, "algorithmic": alg,
                                                       "kernels": [k[:60] for k in fetch if any(p in k for p in PER_READ)]}
    # FASTQ legs (bench.py --fastq): the whole step's traffic, every kernel's mean x dispatches over the number of steps
    # (= dispatches of the pass that reads the text: fq1_scan_kernel or nl_scan_kernel); FETCH_SIZE is doubled for the
    # kernels that stream the text with 16-byte loads (the gfx950 correction), taken as it is for the others
    if line is not None and "results" in line and any("fq1_scan_kernel" in k or "nl_scan_kernel" in k for k in fetch):
        one_pass = any("fq1_scan_kernel" in k for k in fetch)
        general = ("nl_", "line_sum_kernel", "fastq_header_kernel", "fastq_finalize_kernel", "tile_kernel", "dense_kernel")
        streaming = ("fq1_scan_kernel", "nl_scan_kernel", "nl_fill_kernel", "tile_kernel", "dense_kernel", "line_sum_kernel")

        def step_bytes(means, counter_steps):
            # A context's first text (a warm-up step) goes through the general path whatever its lines: with one-pass
            # kernels in the trace those kernels are left out, the one-pass kernels count once per step, the kernels both
            # kinds of step share by their dispatches over all steps of the run.
            tot = 0.0
            for k, v in means.items():
                if "synth" in k or (one_pass and any(p in k for p in general)):
                    continue
                per_step = 1.0 if (one_pass and "fq1_" in k) else v["dispatches"] / counter_steps
                tot += v["mean_KB"] * 1024 * per_step * (2 if (means is fetch and any(p in k for p in streaming)) else 1)
            return tot

        pmc_steps = 3   # tools/profile_round.sh: --steps 2 --warmup 1 in the counter passes
        res["fastq_step_traffic_bytes"] = {"fetch_corrected": step_bytes(fetch, pmc_steps), "write": step_bytes(write, pmc_steps),
                                           "path": "one pass (ppqc_fastq1.hip)" if one_pass else "general (ppqc_fastq_gpu.hip)",
                                           "workload": line["config"]["workload"]}
    res["csrc_sha16"] = sha
    json.dump(res, open(prefix + "_pmc_hbm.json", "w"), indent=1)
    print(json.dumps(res.get("tile_kernel_traffic_bytes_per_launch"), indent=1))
    sq = pmc_rows(src + "/sq")
    if sq:
        out = {"csrc_sha16": sha, "unit": "per-launch mean", "kernels": {}}
        for k, d in sq.items():
            if any(p in k for p in PER_READ) or "tables_kernel" in k or "fq1_" in k or "nl_scan" in k:
                out["kernels"][k[:80]] = {c: v[1] / v[0] for c, v in d.items()}
                out["kernels"][k[:80]]["dispatches"] = max(v[0] for v in d.values())
        json.dump(out, open(prefix + "_pmc_sq.json", "w"), indent=1)
    if ks:
        print(open(ks[0]).read()[:2500])
    if line and "roofline" in line:
        print(json.dumps(line["roofline"]))
    if "fastq_step_traffic_bytes" in res:
        print(json.dumps(res["fastq_step_traffic_bytes"]))


if __name__ == "__main__":
    main()
