#!/usr/bin/env python3
"""How far the host runs ahead of the device: per frame, when submit returned and when collect returned (development tool)."""
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
enc = ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25)
depth = enc.pipeline_depth
N = 150
ts, tc, tsub = [0.0] * N, [0.0] * N, [0.0] * N
q, nc = 0, 0
t00 = time.perf_counter()
for i in range(N):
    if q == depth:
        enc.collect(); tc[nc] = time.perf_counter() - t00; nc += 1; q -= 1
    y, uv = fr[i % 26 if i % 25 else 0]
    enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
    a = time.perf_counter()
    enc.submit()
    ts[i] = time.perf_counter() - t00; tsub[i] = ts[i] - (a - t00); q += 1
while q:
    enc.collect(); tc[nc] = time.perf_counter() - t00; nc += 1; q -= 1
print("depth", depth)
for i in range(95, 135):
    print("frame %3d%s submit at %8.3f ms (call %.3f ms)  collected at %8.3f ms  in flight for %6.3f ms" % (
        i, " IDR" if i % 25 == 0 else "    ", ts[i] * 1e3, tsub[i] * 1e3, tc[i] * 1e3, (tc[i] - ts[i]) * 1e3))
