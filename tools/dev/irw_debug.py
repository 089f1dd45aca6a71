#!/usr/bin/env python3
"""Where k_intra_rows first leaves the oracle: per macroblock and block (GPU box; development)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol  # noqa: E402
import cedarx_h264_encoder_amd as ce  # noqa: E402


def report(tag, oe, ge, ref, got, w):
    om = oe.mb_array()
    gc = ge.debug_read("coef")
    mbw = (w + 15) // 16
    ol_ = om["luma"].reshape(len(om), 16, 16)
    gl = gc[:, 0:256].reshape(len(om), 16, 16)
    bad = [(m, b) for m in range(len(om)) for b in range(16) if not np.array_equal(ol_[m, b], gl[m, b])]
    baddc = [m for m in range(len(om)) if not np.array_equal(om["dc16"][m], gc[m, 256:272])]
    print("%s: bytes %s, luma blocks differing %d, dc16 mbs differing %d" % (tag, ref == got, len(bad), len(baddc)))
    types = om["type"]
    seen = set()
    for m, b in bad:
        if m in seen:
            continue
        seen.add(m)
        x, y = m % mbw, m // mbw
        nb = {k: (int(types[mm]) if 0 <= mm < len(om) else None) for k, mm in
              (("L", m - 1 if x else -1), ("T", m - mbw), ("TL", m - mbw - 1 if x else -1), ("TR", m - mbw + 1 if x < mbw - 1 else -1), ("R", m + 1 if x < mbw - 1 else -1))}
        bl = [bb for mm, bb in bad if mm == m]
        print("   mb x %d y %d type %d bad blocks %s modes %s neighbours %s" % (x, y, types[m], bl, [int(om["i4_modes"][m][bb]) for bb in bl], nb))
        if len(seen) >= 10:
            break
    return len(bad)


w, h, qp = 320, 240, 30
y0, uv0 = ol.gen_frame("B", w, h, 0)
for rep in range(3):
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=1))
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=1) as ge:
        for f in range(3):
            ref = oe.encode(y0, uv0)
            got = ge.encode_frame(y0, uv0)
            report("session %d, frame 0's content as frame %d" % (rep, f), oe, ge, ref, got, w)
for (w, h, qp, pat, n) in ((640, 368, 24, "B", 3), (1920, 1080, 24, "B", 2)):
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=1))
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=1) as ge:
        for f in range(n):
            y, uv = ol.gen_frame(pat, w, h, f)
            ref = oe.encode(y, uv)
            got = ge.encode_frame(y, uv)
            report("%dx%d frame %d" % (w, h, f), oe, ge, ref, got, w)

# origin hunt: macroblocks whose levels all agree but whose (deblocked) reconstruction differs
w, h, qp = 320, 240, 30
oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=1))
with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=1) as ge:
    ref = oe.encode(y0, uv0)
    got = ge.encode_frame(y0, uv0)
    om = oe.mb_array()
    gc = ge.debug_read("coef")
    ry, _ = oe.recon()
    gy = ge.debug_read("recon_y")
    mbw, mbh = w // 16, h // 16
    for my in range(mbh):
        for mx in range(mbw):
            m = my * mbw + mx
            d = ry[16 * my:16 * my + 16, 16 * mx:16 * mx + 16].astype(int) - gy[16 * my:16 * my + 16, 16 * mx:16 * mx + 16].astype(int)
            lev_ok = np.array_equal(om["luma"][m].reshape(-1), gc[m, 0:256]) and np.array_equal(om["dc16"][m], gc[m, 256:272])
            if d.any():
                print("mb x %d y %d type %d i16mode %d levels %s: %d samples differ" % (mx, my, om["type"][m], om["i16_mode"][m], "EQUAL" if lev_ok else "differ", int((d != 0).sum())))
                if lev_ok and om["type"][m] == 1:
                    print(d)
