#!/usr/bin/env python3
"""Development probe: what the sub-sample refinement costs (1080p, 8 lock-step streams) and buys."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cedarx_h264_encoder_amd as ce
W, H, S = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
def tex(f, vx=1.25, vy=0.5):
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    u, v = x + vx * f, y + vy * f
    lum = 128 + 60 * np.sin(u / 5.0) * np.cos(v / 7.0) + 30 * np.sin((u + v) / 3.0)
    yy = np.clip(np.round(lum), 0, 255).astype(np.uint8).reshape(-1)
    uv = np.full((H // 2, W), 128, np.uint8)
    return yy, uv.reshape(-1)
frames = []
for f in range(26):
    y, uv = tex(f)
    frames.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
for sp in (0, 1, 2):
    enc = ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25, n_streams=S, subpel=sp)
    depth = enc.pipeline_depth
    def run(first, n):
        q, nb = 0, 0
        for i in range(first, first + n):
            if q == depth:
                nb += enc.collect(); q -= 1
            for s in range(S):
                y, uv = frames[i % 25 if i % 25 else 0] if False else frames[i % 26]
                enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
            enc.submit(); q += 1
        while q:
            nb += enc.collect(); q -= 1
        return nb
    run(0, 26)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nb = run(26, 100)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("subpel %d: %.3f ms/step, %.1f M MB/s, %.0f bytes per frame and stream" % (sp, dt / 100 * 1e3, S * 8160 * 100 / dt / 1e6, nb / 100.0 / S), flush=True)
    enc.close()
