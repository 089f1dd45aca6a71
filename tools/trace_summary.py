#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of a bench.py run (runs on the GPU box).

    tools/trace_summary.py <kernel_trace.csv> [frames_per_stream_in_steady_state]

Prints, for the session's kernels (cedar::*, config-time probes left out):
  * per kernel: launches, average / min / max duration, share of the busy time;
  * the steady-state window (second half of the longest uninterrupted run of dispatches = the
    measured session's timed region): wall span, the union of
    all kernel intervals ("GPU busy"), the idle gaps between them and, per hardware
    queue, the busy time -- i.e. which kernel or which launch gap owns a step.
The full trace is large; this text is what gets committed under profiles/.
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not name.startswith("cedar::") or "k_probe" in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
    rows.sort()
    if not rows:
        print("no cedar:: kernels in", path)
        return
    print("kernels: %d dispatches" % len(rows))
    agg = collections.defaultdict(list)
    for s, e, n, q in rows:
        agg[n].append((e - s) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print("%-44s %8s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "share"))
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-44s %8d %10.1f %10.1f %10.1f %6.1f%%" % (n[:44], len(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    # bench.py runs several sessions (a throw-away one to warm the runtime, the measured one, one with
    # frames back to back for the work times): they are separated by milliseconds of set-up. Steady
    # state = the second half of the longest run of dispatches without a 1.5 ms hole.
    runs, cur = [], [rows[0]]
    for prev, r in zip(rows, rows[1:]):
        if r[0] - max(prev[1], cur[-1][1]) > 1500000:
            runs.append(cur)
            cur = []
        cur.append(r)
    runs.append(cur)
    ss = max(runs, key=len)
    print("sessions seen: %d (dispatches %s); the longest is summarised" % (len(runs), [len(x) for x in runs]))
    ss = ss[len(ss) // 2:]
    t0, t1 = ss[0][0], max(e for _, e, _, _ in ss)
    span = (t1 - t0) / 1e3
    busy, cur_s, cur_e = 0.0, None, None
    gaps = []
    for s, e, n, q in ss:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += (cur_e - cur_s) / 1e3
                gaps.append(((s - cur_e) / 1e3, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += (cur_e - cur_s) / 1e3
    print("steady state: %d dispatches, span %.1f us, union busy %.1f us (%.1f %%), %d idle gaps, %.1f us idle" % (
        len(ss), span, busy, 100 * busy / span, len(gaps), sum(g for g, _ in gaps)))
    by_next = collections.defaultdict(lambda: [0, 0.0])
    for g, n in gaps:
        by_next[n][0] += 1
        by_next[n][1] += g
    for n, (c, g) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:8]:
        print("  idle before %-40s %6d gaps %10.1f us (avg %.1f)" % (n[:40], c, g, g / c))
    perq = collections.defaultdict(float)
    for s, e, n, q in ss:
        perq[q] += (e - s) / 1e3
    for q, b in sorted(perq.items(), key=lambda kv: -kv[1]):
        print("  queue %-6s busy %10.1f us (%.1f %% of span)" % (q, b, 100 * b / span))
    sagg = collections.defaultdict(float)
    for s, e, n, q in ss:
        sagg[n] += (e - s) / 1e3
    print("steady-state kernel time / span:")
    for n, b in sorted(sagg.items(), key=lambda kv: -kv[1]):
        print("  %-44s %10.1f us  %.3f of span" % (n[:44], b, b / span))


if __name__ == "__main__":
    main()
