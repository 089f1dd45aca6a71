#!/usr/bin/env python3
"""Development: a pipelined 1080p session under the schedule knobs given as KEY=VALUE arguments; says which frame fails."""
import os, sys, time
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cedarx_h264_encoder_amd.encoder as ce
w, h, n = 1920, 1080, 30
frames = [ce.gen_pattern("B", w, h, f) for f in range(n)]
t0 = time.time()
try:
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25) as enc:
        depth, q, got = enc.pipeline_depth, 0, 0
        for f, fr in enumerate(frames):
            if q == depth:
                enc.collect(); got += 1; q -= 1
            enc.set_input(*fr)
            enc.submit(); q += 1
        while q:
            enc.collect(); got += 1; q -= 1
    print(sys.argv[1:], "ok", got, "frames, depth", depth, "%.2f s" % (time.time() - t0))
except Exception as ex:
    print(sys.argv[1:], "FAILED after", got, "collected:", str(ex)[-120:], "%.2f s" % (time.time() - t0))
