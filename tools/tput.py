#!/usr/bin/env python3
"""Throughput of S lock-step 1080p streams with the full submit/collect pipeline, no profiling events (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cedarx_h264_encoder_amd as ce
W, H = int(os.environ.get("TPUT_W", "1920")), int(os.environ.get("TPUT_H", "1080"))
SLICES = int(os.environ.get("TPUT_SLICES", "1"))
QP = int(os.environ.get("TPUT_QP", "24"))
MBS = ((W + 15) // 16) * ((H + 15) // 16)
dev = torch.device("cuda", 0)
frames = []
NSRC = 8 if W * H <= 1920 * 1088 else 2
for s in range(NSRC):
    fr = []
    for f in range(26):
        y, uv = ce.gen_pattern("B", W, H, f, s << 8)
        fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
    frames.append(fr)
torch.cuda.synchronize()
host_frames = [[(y.cpu().numpy(), uv.cpu().numpy()) for y, uv in fr] for fr in frames] if os.environ.get("TPUT_HOST") == "1" else None
for S in [int(x) for x in sys.argv[1:]] or [1]:
    enc = None  # (one pipelined session per process and device: let the previous one go first)
    enc = ce.CedarEncoder(width=W, height=H, qp=QP, keyframe_interval=int(os.environ.get("KEYINT", "25")), n_streams=S, slice_count=SLICES)
    depth = enc.pipeline_depth
    HOST = os.environ.get("TPUT_HOST") == "1"   # frames handed over in the pinned host buffers (the reference contract)
    def feed(i):
        for s in range(S):
            y, uv = frames[s % NSRC][i % 26 if i % 25 else 0]
            if HOST:
                hy, huv = host_frames[s % NSRC][i % 26 if i % 25 else 0]
                enc.set_input(hy, huv, s)   # memcpy into the session's pinned buffers; submit does the H2D
            else:
                enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
        enc.submit()
    q = 0
    for i in range(50):   # pipelined warm-up: the runtime's one-time pool growth stalls land here
        if q == depth:
            enc.collect(); q -= 1
        feed(i); q += 1
    while q:
        enc.collect(); q -= 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n, q = 200, 0
    for i in range(50, 50 + n):
        if q == depth:
            enc.collect(); q -= 1
        feed(i); q += 1
    while q:
        enc.collect(); q -= 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%dx%d slices %d S=%d depth %d: %.3f ms/step, %.2f M MB/s" % (W, H, SLICES, S, depth, dt / n * 1e3, S * MBS * n / dt / 1e6), flush=True)
    enc.close()
