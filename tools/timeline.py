"""usage: python tools/timeline.py <dir with *_kernel_trace.csv> [n_last_kernels]
The last kernels of a rocprofv3 --kernel-trace run as a timeline: start (us, from the first one shown), duration, queue,
the idle gap in front of each on the whole device, and the name.  For reading a step's critical path off a trace."""
import csv, glob, sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
rows = rows[-last:]
t0 = rows[0][0]
busy_to = rows[0][0]
for s, e, q, name in rows:
    gap = (s - busy_to) / 1e3
    short = name.split("(")[0].replace("void ", "").replace("ppqc::", "")
    print("%9.1f  %8.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, ("gap %5.1f  " % gap) if gap > 0 else "           ", short[:70]))
    busy_to = max(busy_to, e)
