#!/usr/bin/env python3
"""Time-ordered kernel launches of the LAST training step in a rocprofv3 kernel trace (csv): start offset, duration, gap to the previous
kernel's end, name.  Usage: step_timeline.py <kernel_trace.csv> [first-kernel-substring=k_pack_views]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_pack_views"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
n_back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lo, hi = idx[-n_back], idx[-n_back + 1] if n_back > 1 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"].split("(")[0][:90]))
    prev_end = max(prev_end, e)
print("step span %.1f us" % ((prev_end - t0) / 1e3))
