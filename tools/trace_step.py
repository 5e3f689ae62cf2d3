"""Kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV (relative to the start of its per-read kernel).
usage: python tools/trace_step.py <dir with *kernel_trace.csv> [per-read kernel name part]"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else None
names = ("tile_kernel", "dense_kernel", "nl_scan") if key is None else (key,)
idx = [i for i, r in enumerate(rows) if any(n in r["Kernel_Name"] for n in names)]
last = idx[-1]
while last - 1 in idx or (last - 1 >= 0 and "extremes" in rows[last - 1]["Kernel_Name"]):   # tile + dense pair
    last -= 1
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[max(0, last - 2):]:
    print("%9.1f %9.1f  %7.1f  %-60s q%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                            r["Kernel_Name"].replace("ppqc::", "").replace("void ", "")[:60], r.get("Queue_Id")))
