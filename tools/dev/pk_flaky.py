#!/usr/bin/env python3
"""Many small sessions against the oracle; where do the bytes differ when they do (GPU box; development)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol  # noqa: E402
import cedarx_h264_encoder_amd as ce  # noqa: E402

bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for (w, h, qp, keyint, n) in ((64, 48, 30, 1, 3), (320, 240, 30, 25, 4)):
        oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint))
        with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint) as ge:
            for f in range(n):
                y, uv = ol.gen_frame("B", w, h, f)
                ref = oe.encode(y, uv)
                got = ge.encode_frame(y, uv)
                if ref != got:
                    bad += 1
                    d = next((i for i in range(min(len(ref), len(got))) if ref[i] != got[i]), min(len(ref), len(got)))
                    nd = sum(1 for i in range(min(len(ref), len(got))) if ref[i] != got[i])
                    print("rep %d %dx%d frame %d: len ref %d got %d, first diff at %d, %d bytes differ; pipelined %s; got[d:d+16]=%s ref=%s" % (
                        rep, w, h, f, len(ref), len(got), d, nd, ge.stats()["frames_pipelined"], got[d:d + 16].hex(), ref[d:d + 16].hex()))
print("sessions with a difference:", bad)
