#!/usr/bin/env python3
"""Development probe: S lock-step 1080p streams as ONE session against the same streams as N
sessions of S/N that one thread feeds in turn (the deblocking of one group then runs beside the
motion search of the next). Usage: groups_probe.py S N [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cedarx_h264_encoder_amd as ce
W, H = 1920, 1080
MBS = 8160
dev = torch.device("cuda", 0)
frames = []
for s in range(8):
    fr = []
    for f in range(26):
        y, uv = ce.gen_pattern("B", W, H, f, s << 8)
        fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
    frames.append(fr)
torch.cuda.synchronize()
S = int(sys.argv[1])
for N in [int(x) for x in sys.argv[2:]]:
    per = S // N
    encs = [ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25, n_streams=per) for _ in range(N)]
    depth = min(e.pipeline_depth for e in encs)
    def feed(i):
        for k, enc in enumerate(encs):
            for s in range(per):
                y, uv = frames[(k * per + s) % 8][i % 26 if i % 25 else 0]
                enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
            enc.submit()
    def take():
        for enc in encs:
            enc.collect()
    def run(first, n):
        q = 0
        for i in range(first, first + n):
            if q == depth:
                take(); q -= 1
            feed(i); q += 1
        while q:
            take(); q -= 1
    run(0, 50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    run(50, n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("S=%d as %d session(s) of %d, depth %d: %.3f ms/step, %.2f M MB/s" % (S, N, per, depth, dt / n * 1e3, S * MBS * n / dt / 1e6), flush=True)
    for enc in encs:
        enc.close()
