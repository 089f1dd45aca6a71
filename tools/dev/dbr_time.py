#!/usr/bin/env python3
"""Development: deblock kernel time per P picture, frames back to back (CEDAR_ROW_PIPE=0), for pictures of 1, 2, 4, 17 bands
of 4 macroblock rows: the cost of a step (one band, no hand-off) and of a band boundary. KEY=VALUE arguments set the environment."""
import os, sys
os.environ["CEDAR_ROW_PIPE"] = "0"
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cedarx_h264_encoder_amd.encoder as ce
for w, h in ((1920, 64), (1920, 128), (1920, 256), (1920, 1088), (3840, 64)):
    frames = [ce.gen_pattern("B", w, h, f) for f in range(13)]
    with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=31) as enc:
        for f in range(3):
            enc.encode_frame(*frames[f])
        enc.reset_stats()
        enc.set_profiling(True, stages=["deblock"])
        for f in range(3, 13):
            enc.encode_frame(*frames[f])
        st = enc.stats()
    ms = st["ms"]["deblock"] / max(1, st["launches"]["deblock"])
    steps = w // 16 + 3
    print("%4dx%-4d bands %2d: deblock %.1f us per picture, %.2f us per step of the first band" % (w, h, (h // 16 + 3) // 4, ms * 1e3, ms * 1e3 / steps), sys.argv[1:])
