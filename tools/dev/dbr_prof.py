#!/usr/bin/env python3
"""Development (with the cycle counters compiled into enc_k_deblock.hip): one band, a few P pictures back to back."""
import os, sys
os.environ["CEDAR_ROW_PIPE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cedarx_h264_encoder_amd.encoder as ce
w, h = 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = [ce.gen_pattern("B", w, h, f) for f in range(4)]
with ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=31) as enc:
    for f in range(4):
        enc.encode_frame(*frames[f])
