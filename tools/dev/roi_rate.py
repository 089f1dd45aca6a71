"""What a session with per-macroblock QPs costs in throughput (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cedarx_h264_encoder_amd as ce
W, H, K = 1920, 1080, 25
dev = torch.device("cuda", 0)
frames = []
for f in range(50):
    y, uv = ce.gen_pattern("B", W, H, f, 0)
    frames.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
ROI = dict(roi=((40, 20, 40, 30, -8), (0, 60, 120, 8, 12)))
MAD = dict(mad_th=(1, 3, 6, 12), mad_dqp=(-3, -1, 0, 2, 4))
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, kw, n in (("plain", {}, 300), ("roi", ROI, 300), ("mad", MAD, 300), ("refresh 30", dict(intra_refresh=30), 300),
                    ("plain, 8 GOPs side by side", dict(gop_parallel=8), 800), ("roi + mad, 8 GOPs side by side", dict(gop_parallel=8, **ROI, **MAD), 800),
                    ("refresh 30, 8 GOPs side by side", dict(gop_parallel=8, intra_refresh=30), 800)):
    if only not in name:
        continue
    with ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=K, **kw) as enc:
        depth, q, nbytes = enc.pipeline_depth, 0, 0
        def take():
            global nbytes
            enc.collect(); nbytes += len(enc.output(0))
        for warm in ((False,) if "gop_parallel" in kw else (True, False)):   # (a GOP-lane session ends with its flush)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(n):
                if q == depth:
                    take(); q -= 1
                y, uv = frames[i % 50]
                enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
                enc.submit(); q += 1
            enc.flush()
            while q:
                take(); q -= 1
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if warm:
                nbytes = 0
    print("%-34s %.3f ms/frame  %.1f M MB/s  %.1f KB/frame" % (name, dt / n * 1e3, n * 8160 / dt / 1e6, nbytes / n / 1e3), flush=True)
