#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag>/ (tools/collect_r04.sh) into profiles/r04_*."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04c"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for leg in ("w1", "p1", "s8", "s32"):
    for suffix in ("_kernel_stats.csv", "_trace_summary.txt", ".json"):
        f = os.path.join(src, leg + suffix)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, "r04_%s%s" % (leg, "_bench.json" if suffix == ".json" else suffix)))
for name, geom in (("s1", "1920x1080x1"), ("s32", "1920x1080x32")):
    f = os.path.join(src, "pmc_%s_FETCH_SIZE.csv" % name)
    w = os.path.join(src, "pmc_%s_WRITE_SIZE.csv" % name)
    if os.path.exists(f) and os.path.exists(w):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f, w, "r04_" + name, geom])
print("published", tag)
