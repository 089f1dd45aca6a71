#!/usr/bin/env python3
"""Instruction counts of one kernel of a hipcc -S listing (compiled with -gline-tables-only), per source-line range:
    python tools/dev/isa_buckets.py file.s <mangled kernel name> "[(first, last, \"name\"), ...]"
VALU slots count a quarter-rate v_qsad_pk_u16_u8 as 4. Development tool behind DESIGN.md 5.5 and the k_inter line of bench.py."""
import re, sys, collections
s = open(sys.argv[1]).read().split("\n")
fn = sys.argv[2]
# find function start
start = next(i for i, l in enumerate(s) if l.startswith(fn + ":"))
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
files = {}
for l in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = (0, 0)
cnt = collections.Counter(); cntv = collections.Counter()
for l in s[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'\s+([a-z_0-9]+)\s', l + " ")
    if m and not l.strip().startswith(".") and not l.strip().startswith(";"):
        op = m.group(1)
        f = files.get(cur[0], "?").split("/")[-1]
        key = (f, cur[1])
        cnt[key] += 1
        if op.startswith("v_"): cntv[key] += (4 if "qsad" in op else 1)
ranges = eval(sys.argv[3]) if len(sys.argv) > 3 else None
if ranges:
    tot = collections.Counter(); totv = collections.Counter()
    for (f, ln), c in cnt.items():
        name = "other:" + f
        if f.endswith("enc_k_parallel.hip"):
            for (a, b, n) in ranges:
                if a <= ln <= b: name = n; break
            else: name = "unranged"
        tot[name] += c; totv[name] += cntv[(f, ln)]
    for n in sorted(tot, key=lambda n: -totv[n]): print("%-28s all %5d  valu-slots %5d" % (n, tot[n], totv[n]))
    print("total", sum(tot.values()), sum(totv.values()))
else:
    for k in sorted(cnt): print(k, cnt[k], cntv[k])
