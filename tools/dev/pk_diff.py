#!/usr/bin/env python3
"""Pipelined sessions against the oracle: WHERE an access unit differs when it does (GPU box; development)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol  # noqa: E402
import cedarx_h264_encoder_amd as ce  # noqa: E402

w, h, n, keyint, qp = 640, 368, 120, 15, 29
frames = [ol.gen_frame("BC"[f % 2], w, h, f) for f in range(n)]
oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint))
want = [oe.encode(y, uv) for y, uv in frames]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint) as enc:
        got, q = [], 0
        for y, uv in frames:
            if q == enc.pipeline_depth:
                enc.collect(); got.append(enc.output(0)); q -= 1
            enc.set_input(y, uv)
            enc.submit(); q += 1
        while q:
            enc.collect(); got.append(enc.output(0)); q -= 1
    bad = [f for f in range(n) if got[f] != want[f]]
    print("rep %d: differing frames %s" % (rep, bad))
    for f in bad[:6]:
        a, b = np.frombuffer(got[f], np.uint8), np.frombuffer(want[f], np.uint8)
        m = min(a.size, b.size)
        d = np.nonzero(a[:m] != b[:m])[0]
        print("   frame %d (%s): len got %d want %d; %d bytes differ, first %d last %d; got %s want %s" % (
            f, "IDR" if f % keyint == 0 else "P", a.size, b.size, d.size, d[0] if d.size else -1, d[-1] if d.size else -1,
            bytes(a[d[0]:d[0] + 12]).hex() if d.size else "", bytes(b[d[0]:d[0] + 12]).hex() if d.size else ""))
        # is it an earlier/later frame's content?
        for g in range(max(0, f - 20), min(n, f + 20)):
            if g != f and d.size and want[g][d[0]:d[0] + 12] == bytes(a[d[0]:d[0] + 12]):
                print("      ... those bytes are frame %d's at the same offset" % g)
