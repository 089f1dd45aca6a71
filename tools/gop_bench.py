import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
import cedarx_h264_encoder_amd as ce
W, H, K = 1920, 1080, 25
dev = torch.device("cuda", 0)
frames = []
for f in range(50):
    y, uv = ce.gen_pattern("B", W, H, f, 0)
    frames.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
for G in [int(x) for x in sys.argv[1:]]:
    n = 4 * G * K
    with ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=K, gop_parallel=G) as enc:
        depth, q = enc.pipeline_depth, 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            if q == depth:
                enc.collect(); q -= 1
            y, uv = frames[i % 50]
            enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
            enc.submit(); q += 1
        enc.flush()
        while q:
            enc.collect(); q -= 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("G=%d depth %d: %d frames, %.3f ms/frame, %.1f M MB/s" % (G, depth, n, dt / n * 1e3, n * 8160 / dt / 1e6), flush=True)
