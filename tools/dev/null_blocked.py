"""One 1080p stream pipelined while ANOTHER queue of the process holds a running kernel with packets behind it: PyTorch's
default (null) stream, whose hardware queue is the pipe-mate of the session's analysis stream when the session's seven
streams and the null stream fill the eight hardware queues in order (development aid: the slow mode of DESIGN 5.9)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cedarx_h264_encoder_amd as ce
W, H, K = 1920, 1080, 25
block = len(sys.argv) > 1 and sys.argv[1] == "block"
dev = torch.device("cuda", 0)
frames = []
for f in range(50):
    y, uv = ce.gen_pattern("B", W, H, f, 0)
    frames.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
with ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=K) as enc:
    depth, q = enc.pipeline_depth, 0
    def run(n):
        global q
        t0 = time.perf_counter()
        for i in range(n):
            if q == depth:
                enc.collect(); q -= 1
            y, uv = frames[i % 50]
            enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
            enc.submit(); q += 1
        while q:
            enc.collect(); q -= 1
        return n * (W // 16) * ((H + 15) // 16) / (time.perf_counter() - t0) / 1e6
    run(100)
    a = run(600)
    if block:   # ~0.4 s of spinning one-thread kernels on the null stream, one behind the other
        for _ in range(20):
            torch.cuda._sleep(40_000_000)
    b = run(600)
    torch.cuda.synchronize()
    c = run(600)
print("priorities %s: %.1f M MB/s, %s %.1f M, afterwards %.1f M" % (os.environ.get("CEDAR_STREAM_PRIO", "default"), a,
      "with spinning kernels queued on the null stream" if block else "again", b, c))
