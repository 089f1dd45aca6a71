"""Host time of cedar_enc_submit per frame against the device's frame period (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cedarx_h264_encoder_amd as ce
W, H, K = 1920, 1080, 25
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # lock-step streams
dev = torch.device("cuda", 0)
frames = []
for f in range(50):
    y, uv = ce.gen_pattern("B", W, H, f, 0)
    frames.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
with ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=K, n_streams=S) as enc:
    depth, q = enc.pipeline_depth, 0
    ts, tc, ti = [], [], []
    t_all = time.perf_counter()
    for i in range(600):
        if q == depth:
            t0 = time.perf_counter(); enc.collect(); tc.append(time.perf_counter() - t0); q -= 1
        t0 = time.perf_counter()
        for s_ in range(S):
            y, uv = frames[(i + 3 * s_) % 50]
            enc.set_device_input(s_, y.data_ptr(), uv.data_ptr())
        ti.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); enc.submit(); ts.append(time.perf_counter() - t0); q += 1
    while q:
        enc.collect(); q -= 1
    dt = time.perf_counter() - t_all
ts = sorted(ts[50:]); tc = sorted(tc[50:]); ti = sorted(ti[50:])
print("%d stream(s), depth %d: set_device_input x%d: median %.1f us; submit: median %.1f us, p90 %.1f us; collect (blocking): median %.1f us; frame period %.1f us" % (
    S, depth, S, ti[len(ti) // 2] * 1e6, ts[len(ts) // 2] * 1e6, ts[len(ts) * 9 // 10] * 1e6, tc[len(tc) // 2] * 1e6, dt / 600 * 1e6))
