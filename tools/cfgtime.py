#!/usr/bin/env python3
"""How long cedar_enc_config takes, stream probes included (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cedarx_h264_encoder_amd as ce
for w, h, s in ((1920, 1080, 1), (1920, 1080, 1), (3840, 2160, 1), (1920, 1080, 32)):
    t0 = time.perf_counter()
    e = ce.CedarEncoder(width=w, height=h, n_streams=s)
    t1 = time.perf_counter()
    e.close()
    print("%dx%d x%d: config %.0f ms, close %.0f ms" % (w, h, s, (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
