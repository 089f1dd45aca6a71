#!/usr/bin/env python3
"""Per-stage GPU time for S lock-step 1080p streams (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cedarx_h264_encoder_amd as ce
W, H = 1920, 1080
dev = torch.device("cuda", 0)
frames = []
for s in range(8):
    fr = []
    for f in range(26):
        y, uv = ce.gen_pattern("B", W, H, f, s << 8)
        fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
    frames.append(fr)
torch.cuda.synchronize()
for S in [int(x) for x in sys.argv[1:]] or [1, 8, 32]:
    enc = ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25, n_streams=S)
    def feed(i):
        for s in range(S):
            y, uv = frames[s % 8][i % 26 if i % 25 else 0]
            enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
        enc.submit()
    for i in range(3):
        feed(i); enc.collect()
    enc.set_profiling(True); enc.reset_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    feed(3)
    for i in range(4, 3 + n):
        feed(i); enc.collect()
    enc.collect()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = enc.stats()
    per = {k: round(st["ms"][k] / max(1, st["launches"][k]), 3) for k in st["ms"]}
    print("S=%d: %.2f ms/step, %.2f M MB/s, per-launch ms %s" % (S, dt / n * 1e3, S * 8160 * n / dt / 1e6, per), flush=True)
    enc.close()
