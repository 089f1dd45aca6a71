#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/."""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, geom = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "1920x1080x1")
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0], os.path.join(dst, tag + "_rocprofv3_kernel_stats.csv"))
f = glob.glob(os.path.join(src, "pmc_FETCH_SIZE", "*", "*counter_collection.csv"))[0]
w = glob.glob(os.path.join(src, "pmc_WRITE_SIZE", "*", "*counter_collection.csv"))[0]
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f, w, tag, geom])
