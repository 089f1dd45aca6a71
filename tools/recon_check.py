#!/usr/bin/env python3
"""Where does the GPU reconstruction differ from the oracle? (1080p, two frames; development tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import oracle_lib as ol
import cedarx_h264_encoder_amd as ce
w, h = 1920, 1080
oe = ol.OracleEncoder(ol.default_config(w, h, qp=26, keyint=25))
with ce.CedarEncoder(width=w, height=h, qp=26, keyframe_interval=25) as enc:
    for f in range(2):
        y, uv = ol.gen_frame("B", w, h, f)
        a = oe.encode(y, uv); b = enc.encode_frame(y, uv)
        ry, ruv = oe.recon()
        gy = enc.debug_read("recon_y"); guv = enc.debug_read("recon_uv")
        dy = np.argwhere(gy != ry); duv = np.argwhere(guv != ruv)
        print("frame", f, "bytes", a == b, "luma diffs", len(dy), "chroma diffs", len(duv))
        if len(dy):
            rows = sorted(set(int(r) // 16 for r, c in dy)); cols = sorted(set(int(c) // 16 for r, c in dy))
            print("  luma MB rows", rows[:20], "cols", cols[:12], "... first", dy[0], "px rows", sorted(set(int(r) for r, c in dy))[:24])
        if len(duv):
            print("  chroma MB rows", sorted(set(int(r) // 8 for r, c in duv))[:20], "first", duv[0])
