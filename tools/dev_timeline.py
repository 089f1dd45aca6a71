#!/usr/bin/env python3
"""Device timeline of the stages of consecutive frames from the library's own HIP events
(CEDAR_TRACE_TIMELINE=1; no profiler attached, so the host keeps its lead). Development tool."""
import sys, os, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r)
import torch
import cedarx_h264_encoder_amd as ce
W, H = 1920, 1080
dev = torch.device("cuda", 0)
fr = []
for f in range(26):
    y, uv = ce.gen_pattern("B", W, H, f, 0)
    fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
enc = ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25)
enc.set_profiling(True)
depth = enc.pipeline_depth
q = 0
for i in range(150):
    if q == depth:
        enc.collect(); q -= 1
    y, uv = fr[i %% 26 if i %% 25 else 0]
    enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
    enc.submit(); q += 1
while q:
    enc.collect(); q -= 1
''' % ROOT
env = dict(os.environ, CEDAR_TRACE_TIMELINE="1")
r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env)
ev = collections.defaultdict(dict)
for l in r.stderr.splitlines():
    if l.startswith("timeline frame"):
        _, _, fno, st, a, b = l.split()
        ev[int(fno)][st] = (float(a), float(b))
if not ev:
    sys.exit(r.stderr[-2000:])
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (98, 132)
t0 = ev[lo]["input"][0]
prev_end = None
for f in range(lo, hi):
    d = ev[f]
    parts = []
    for st in ("input", "intra_decide", "intra_recon", "inter", "deblock", "entropy"):
        if st in d:
            parts.append("%s %7.3f..%7.3f" % (st[:6], d[st][0] - t0, d[st][1] - t0))
    end = max(v[1] for v in d.values()) - t0
    print("frame %3d%s %s | done %7.3f%s" % (f, " IDR" if f % 25 == 0 else "    ", "  ".join(parts), end,
                                          "" if prev_end is None else " (+%.3f)" % (end - prev_end)))
    prev_end = end
