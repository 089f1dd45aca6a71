#!/usr/bin/env python3
"""Rate / distortion of the round-2 quality features on BASELINE configs[1] (854x480, QP 30, 50 frames,
pattern-B), CPU only: the oracle's bytes are the HIP path's. Prints a markdown table."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                      # noqa: E402
from tests import oracle_lib as ol      # noqa: E402

W, H, QP, K, N = 854, 480, 30, 25, 50
ROI = (16, 8, 20, 14)                   # macroblocks: the middle of the picture


def psnr(sse, n):
    return 10 * math.log10(255.0 ** 2 * n / sse) if sse else 99.0


def run(**kw):
    oe = ol.OracleEncoder(ol.default_config(W, H, qp=QP, keyint=K, **kw))
    nbytes, sse, sse_roi, intra_p = 0, 0.0, 0.0, 0
    x0, y0, x1, y1 = 16 * ROI[0], 16 * ROI[1], 16 * (ROI[0] + ROI[2]), 16 * (ROI[1] + ROI[3])
    for f in range(N):
        y, uv = ol.gen_frame("B", W, H, f)
        nbytes += len(oe.encode(y, uv))
        ry, _ = oe.recon()
        d = (ry[:H, :W].astype(np.int32) - y.reshape(H, W).astype(np.int32)) ** 2
        sse += float(d.sum())
        sse_roi += float(d[y0:y1, x0:x1].sum())
        if f % K:
            intra_p += int((oe.mb_array()["type"] <= ol.MB_I16x16).sum())
    return nbytes / N, psnr(sse, N * W * H), psnr(sse_roi, N * (x1 - x0) * (y1 - y0)), intra_p / (N - N // K)


def main():
    rows = [("fixed QP 30 (the reference's only mode)", {}),
            ("`intra_refresh` 24 (every macroblock renewed within a GOP)", dict(intra_refresh=24)),
            ("`roi`: the middle 20 x 14 macroblocks 6 QP steps finer", dict(roi=[ROI + (-6,)])),
            ("`roi`: the same finer, everything else 4 coarser", dict(roi=[ROI + (-6,), (0, 0, 54, 30, 4)])),
            ("`mad_dqp` -3,-1,0,+2,+4 over thresholds 1,3,6,12", dict(mad_th=[1, 3, 6, 12], mad_dqp=[-3, -1, 0, 2, 4])),
            ("`subpel` 2", dict(subpel=2))]
    print("| 854x480, QP 30, 50 frames pattern-B | bytes/frame | Mbit/s @30 | PSNR Y, picture | PSNR Y, the region | intra macroblocks per P picture |")
    print("|---|---|---|---|---|---|")
    for name, kw in rows:
        b, p, pr, ip = run(**kw)
        print("| %s | %.0f | %.2f | %.2f | %.2f | %.0f |" % (name, b, b * 8 * 30 / 1e6, p, pr, ip), flush=True)


if __name__ == "__main__":
    main()
