#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel trace: every launch of a step (steps end with fit_stages_k) with
its start offset, duration and the idle time in front of it.  usage: step_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]
ends = [i for i, r in enumerate(rows) if "fit_stages_k" in r["Kernel_Name"]]
for s in range(max(1, len(ends) - 2), len(ends)):
    a, b = ends[s - 1] + 1, ends[s]
    t0 = int(rows[ends[s - 1]]["End_Timestamp"])
    print("step ending at launch %d: %d launches, %.3f ms from the end of the previous fit kernel to the end of this one" %
          (b, b - a + 1, (int(rows[b]["End_Timestamp"]) - t0) / 1e6))
    prev_end = t0
    busy = 0
    for r in rows[a:b + 1]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("   +%8.1f us  gap %7.1f us  dur %8.1f us  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, short(r["Kernel_Name"])))
        prev_end = max(prev_end, en)
        busy += en - st
    print("   sum of durations %.3f ms" % (busy / 1e6))
