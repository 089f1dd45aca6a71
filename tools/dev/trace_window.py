#!/usr/bin/env python3
"""A window of a rocprofv3 kernel trace as a per-queue timeline: python trace_window.py kernel_trace.csv [t0_ms] [len_ms].
Development tool (which kernel ran beside which, on which hardware queue, in the steady state of a batch)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n): return n.replace("cedar::", "").replace("void ", "").split("(")[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows))
t_first, t_last = ev[0][0], ev[-1][1]
if len(sys.argv) > 2 and sys.argv[2].startswith("@"):  # @kernel:n = the start of the n-th launch of that kernel
    name, n = sys.argv[2][1:].split(":")
    hits = [s for s, e, k, q in ev if k.startswith(name)]
    sys.argv[2] = str((hits[min(int(n), len(hits) - 1)] - t_first) / 1e6)
a0 = float(sys.argv[2]) if len(sys.argv) > 2 else (t_last - t_first) / 2e6
t0 = (t_last if a0 < 0 else t_first) + int(a0 * 1e6)  # (negative: from the end of the trace)
t1 = t0 + int((float(sys.argv[3]) if len(sys.argv) > 3 else 4.0) * 1e6)
for s, e, n, q in ev:
    if e < t0 or s > t1: continue
    print("q%-3s %9.1f .. %9.1f  %7.1f us  %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
