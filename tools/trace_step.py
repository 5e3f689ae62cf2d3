#!/usr/bin/env python
"""Print the kernel timeline of the last step found in a rocprofv3 kernel trace, starting at the last launch of a
kernel whose name contains `anchor`.  usage: python tools/trace_step.py <dir> <anchor> [n_rows]"""
import csv
import glob
import sys

d, anchor = sys.argv[1], sys.argv[2]
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
mx = max(int(rows[i]["Grid_Size_X"]) for i in idx)
first = [i for i in idx if int(rows[i]["Grid_Size_X"]) == mx][-1]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + nrows]:
    print("%9.1f us +%8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
