#!/usr/bin/env python3
"""k_intra_rows: every Intra16x16 macroblock of a picture, by mode, row-in-band and column, equal or not (development)."""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol  # noqa: E402
import cedarx_h264_encoder_amd as ce  # noqa: E402

for (w, h, qp) in ((320, 240, 30), (1920, 1080, 24)):
    y0, uv0 = ol.gen_frame("B", w, h, 0)
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=1))
    with ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=1) as ge:
        ref = oe.encode(y0, uv0)
        got = ge.encode_frame(y0, uv0)
        om = oe.mb_array()
        gc = ge.debug_read("coef")
        mbw = (w + 15) // 16
        tab = Counter()
        shown = 0
        for m in range(len(om)):
            if om["type"][m] != 1:
                continue
            x, y = m % mbw, m // mbw
            ok = np.array_equal(om["luma"][m].reshape(-1), gc[m, 0:256]) and np.array_equal(om["dc16"][m], gc[m, 256:272])
            # is any neighbour it predicts from wrong already? (then a difference is inherited)
            def lev_ok(mm):
                return np.array_equal(om["luma"][mm].reshape(-1), gc[mm, 0:256]) and np.array_equal(om["dc16"][mm], gc[mm, 256:272])
            nb_ok = all(lev_ok(mm) for mm in ([m - 1] if x else []) + ([m - mbw] if y else []) + ([m - mbw - 1] if x and y else []))
            tab[(int(om["i16_mode"][m]), y & 3, "ok" if ok else ("BAD" if nb_ok else "bad-inherited?"))] += 1
            if not ok and nb_ok and shown < 6:
                shown += 1
                print("  %dx%d mb x %d y %d mode %d types L %d T %d TL %d: dc16 oracle %s gpu %s" % (w, h, x, y, om["i16_mode"][m], om["type"][m - 1], om["type"][m - mbw], om["type"][m - mbw - 1],
                      om["dc16"][m].tolist(), gc[m, 256:272].tolist()))
                bl = [b for b in range(16) if not np.array_equal(om["luma"][m][b], gc[m, 16 * b:16 * b + 16])]
                for b in bl[:3]:
                    print("     block %d oracle %s gpu %s" % (b, om["luma"][m][b].tolist(), gc[m, 16 * b:16 * b + 16].tolist()))
        print("%dx%d: (mode, row in band, verdict) -> count" % (w, h))
        for k in sorted(tab):
            print("   ", k, tab[k])
