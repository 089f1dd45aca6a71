#!/usr/bin/env python3
"""Host-side timestamps of submit / collect for a 1080p stream (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cedarx_h264_encoder_amd as ce
W, H = 1920, 1080
dev = torch.device("cuda", 0)
fr = []
for f in range(26):
    y, uv = ce.gen_pattern("B", W, H, f, 0)
    fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
enc = ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25, n_streams=1)
depth = enc.pipeline_depth
log = []
def feed(i):
    y, uv = fr[i % 26 if i % 25 else 0]
    enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
    t = time.perf_counter(); enc.submit(); log.append(("submit", i, t, time.perf_counter()))
for i in range(25):
    feed(i); enc.collect()
log.clear()
q = 0; t0 = time.perf_counter()
for i in range(25, 325):
    if q == depth:
        t = time.perf_counter(); enc.collect(); log.append(("collect", i - depth, t, time.perf_counter())); q -= 1
    feed(i); q += 1
while q:
    enc.collect(); q -= 1
tail = [x for x in log if x[1] >= 200]
for k, i, a, b in tail[:36]:
    print("%-8s frame %3d  at %8.1f us  took %7.1f us" % (k, i, (a - t0) * 1e6, (b - a) * 1e6))
sub = [b - a for k, i, a, b in tail if k == "submit"]
col = [b - a for k, i, a, b in tail if k == "collect"]
print("submit: n=%d mean %.1f us max %.1f us; collect: mean %.1f us max %.1f us; span %.3f ms/frame" % (
    len(sub), 1e6 * sum(sub) / len(sub), 1e6 * max(sub), 1e6 * sum(col) / len(col), 1e6 * max(col),
    1e3 * (tail[-1][3] - tail[0][2]) / len(sub)))
print("slow calls (> 1.5 ms):")
for k, i, a, b in log:
    if b - a > 1.5e-3:
        print("  %-8s frame %3d took %8.1f us" % (k, i, (b - a) * 1e6))
