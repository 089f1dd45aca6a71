#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV: when the IDR picture's kernels ran relative to the deblock
kernels of the P pictures around it, and the P period while an intra kernel is / is not resident
(development tool)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Stream_Id", "?")))
rows.sort()
intra = [r for r in rows if "k_intra_recon" in r[2]]
db = [r for r in rows if "k_deblock2" in r[2]]
if len(intra) < 4:
    sys.exit("too few IDR pictures in the trace")
I = intra[len(intra) // 2]          # a steady-state IDR
Iprev, Inext = intra[len(intra) // 2 - 1], intra[len(intra) // 2 + 1]
t0 = I[0]
print("intra_recon: start 0, end %.3f ms (stream %s); previous IDR started %.3f ms earlier, next %.3f ms later" % (
    (I[1] - t0) / 1e6, I[3], (t0 - Iprev[0]) / 1e6, (Inext[0] - t0) / 1e6))
win = [r for r in db if Iprev[1] < r[1] and r[0] < Inext[0]]
print("deblock kernels between the previous IDR's end and the next IDR's start: %d" % len(win))
prev_end = None
for s, e, n, st in win:
    gap = "" if prev_end is None else " (+%.3f since previous end)" % ((e - prev_end) / 1e6)
    print("  stream %s start %+8.3f end %+8.3f dur %.3f%s" % (st, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, gap))
    prev_end = e
# everything on the streams the IDR picture used (its analysis, its deblock, the input stream), 6 ms before to 1 ms after
inp = [r for r in rows if "k_input" in r[2]]
streams = {I[3], inp[0][3]} | {r[3] for r in db if r[0] < I[0] < r[1] or abs(r[0] - I[0]) < 200000}
print("kernels on streams %s around the IDR:" % sorted(streams))
for s, e, n, st in rows:
    if st in streams and t0 - 6000000 < s < t0 + 1000000 and "k_deblock2" not in n[:10] or (st in streams and "k_deblock2" in n and abs(s - t0) < 300000):
        print("  stream %s %-28s start %+8.3f end %+8.3f" % (st, n[:28], (s - t0) / 1e6, (e - t0) / 1e6))
# durations while an intra kernel is resident / is not
def resident(t):
    return any(a <= t <= b for a, b, *_ in intra)
import collections
acc = collections.defaultdict(lambda: [0, 0, 0, 0])
for s, e, n, st in rows:
    if s < intra[1][0]:
        continue
    k = n.split("::")[-1][:24]
    i = 0 if resident(s) else 2
    acc[k][i] += e - s; acc[k][i + 1] += 1
print("%-26s %22s %22s" % ("kernel", "intra resident: n, avg us", "not resident: n, avg us"))
for k, (a, an, b, bn) in sorted(acc.items()):
    print("%-26s %8d %12.1f %8d %12.1f" % (k, an, a / an / 1e3 if an else 0, bn, b / bn / 1e3 if bn else 0))
