#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: per-frame busy/idle on the device (development tool)."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], "s%s/q%s" % (r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))))
rows.sort()
# steady state: last 40% of the trace
t0 = rows[int(len(rows) * 0.6)][0]
rows = [r for r in rows if r[0] >= t0]
span = rows[-1][1] - rows[0][0]
# union of busy intervals
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, *_ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.3f ms, device busy %.3f ms (%.1f%%), idle %.3f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in rows:
    k = n.split("(")[0][:60]
    agg[k][0] += e - s; agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%-62s n=%5d total %.3f ms avg %.1f us" % (k, c, t / 1e6, t / c / 1e3))
# print a few frames' worth of launches in order
names = [r for r in rows if "k_input" in r[2]]
if len(names) > 6:
    a, b = names[2][0], names[5][0]
    print("three frames (%.3f ms):" % ((b - a) / 1e6))
    for s, e, n, q in rows:
        if a <= s < b and ((e - s) > 8000 or "k_input" in n or "fill" in n):
            print("  +%8.1f us  %7.1f us  q=%s  %s" % ((s - a) / 1e3, (e - s) / 1e3, q, n.split("(")[0][:50]))
# compact view: k_input / deblock / first and last k_inter per frame, for 8 frames
ins = [r for r in rows if "k_input" in r[2]]
if len(ins) > 12:
    a, b = ins[2][0], ins[11][0]
    print("nine frames (%.3f ms):" % ((b - a) / 1e6))
    for s_, e_, n_, q_ in rows:
        if a <= s_ < b and ("k_input" in n_ or "deblock" in n_ or "intra" in n_ or ("k_gate" in n_ and e_ - s_ > 30000)):
            print("  +%8.1f .. %8.1f us  %-12s %s" % ((s_ - a) / 1e3, (e_ - a) / 1e3, q_, n_.split("(")[0][:40]))
# around the first IDR of the steady-state window
idr = [r for r in rows if "intra_recon" in r[2]]
if idr:
    c = idr[0][0]
    print("around an IDR:")
    for s_, e_, n_, q_ in rows:
        if c - 1.2e6 <= s_ < c + 5.5e6 and ("k_input" in n_ or "deblock" in n_ or "intra" in n_):
            print("  %+9.1f .. %+9.1f us  %-10s %s" % ((s_ - c) / 1e3, (e_ - c) / 1e3, q_, n_.split("(")[0][:36]))
