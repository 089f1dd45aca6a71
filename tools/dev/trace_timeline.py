#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel trace as a timeline (development aid):
   trace_timeline.py <kernel_trace.csv> [n_dispatches] [min_us]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    if not name.startswith("cedar::") or "k_probe" in name:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.replace("cedar::", ""), r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0
runs, cur = [], [rows[0]]
for prev, r in zip(rows, rows[1:]):
    if r[0] - max(prev[1], cur[-1][1]) > 1500000:
        runs.append(cur); cur = []
    cur.append(r)
runs.append(cur)
ss = max(runs, key=len)
mid = ss[len(ss) * 2 // 3:]
t0 = mid[0][0]
k = 0
for s, e, nme, q, g, wg in mid:
    if (e - s) / 1e3 < min_us:
        continue
    print("%9.1f %9.1f  %8.1f us  q%-3s %-28s grid %s wg %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, nme[:28], g, wg))
    k += 1
    if k >= n:
        break
