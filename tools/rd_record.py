#!/usr/bin/env python3
"""Rate / distortion of the encoder policy on the BASELINE configs (CPU only: the C oracle emits
the same bytes as the HIP path, which the parity tests assert): bytes per frame, bitrate at 30
frames/s and PSNR of the decoded pictures against the source, per plane, over the cropped area.
Second column set: the same with intra modes decided on RECONSTRUCTED neighbours (the oracle's
H264O_CLOSED_LOOP_INTRA experiment) -- what the macroblock-parallel open-loop decision costs.

    python tools/rd_record.py [config ...]      (configs: 0 1 2 3; default 0 1 2)
Prints a markdown table."""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "0": ("configs[0] 320x240 I only", 320, 240, 30, 1, 10, 1),
    "1": ("configs[1] 854x480 I+P", 854, 480, 30, 25, 50, 1),
    "2": ("configs[2] 1920x1080 I+P", 1920, 1080, 24, 25, 50, 1),
    "3": ("configs[3] 3840x2160 I+P, 4 slices", 3840, 2160, 24, 25, 25, 4),
}


def run(w, h, qp, keyint, frames, slices):
    import numpy as np
    from tests import oracle_lib as ol
    oe = ol.OracleEncoder(ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices,
                                            level_idc=51 if w * h > 1920 * 1088 else 41))
    sse = [0.0, 0.0, 0.0]
    nbytes = 0
    ibytes = pbytes = ni = 0
    for f in range(frames):
        y, uv = ol.gen_frame("B", w, h, f)
        au = oe.encode(y, uv)
        nbytes += len(au)
        if f % keyint == 0:
            ibytes += len(au)
            ni += 1
        else:
            pbytes += len(au)
        ry, ruv = oe.recon()   # == what a decoder makes of the stream (tests/test_oracle_roundtrip.py, tests/indep)
        sy = y.reshape(h, w).astype(np.int32)
        suv = uv.reshape(h // 2, w).astype(np.int32)
        d = ry[:h, :w].astype(np.int32) - sy
        sse[0] += float((d * d).sum())
        dc = ruv[:h // 2, :w].astype(np.int32) - suv
        sse[1] += float((dc[:, 0::2] ** 2).sum())
        sse[2] += float((dc[:, 1::2] ** 2).sum())
    n = [frames * w * h, frames * w * h / 4, frames * w * h / 4]
    psnr = [10 * math.log10(255.0 ** 2 * n[i] / sse[i]) if sse[i] else 99.0 for i in range(3)]
    return nbytes / frames, ibytes / max(1, ni), pbytes / max(1, frames - ni), psnr


def main():
    if os.environ.get("RD_CHILD"):
        name, w, h, qp, keyint, frames, slices = CONFIGS[sys.argv[1]]
        b, bi, bp, ps = run(w, h, qp, keyint, frames, slices)
        print("%f %f %f %f %f %f" % (b, bi, bp, ps[0], ps[1], ps[2]))
        return
    which = sys.argv[1:] or ["0", "1", "2"]
    print("| config | QP | bytes/frame (I / P) | Mbit/s @30 | PSNR Y / Cb / Cr (dB) | closed-loop intra: bytes/frame, PSNR Y | open-loop cost |")
    print("|---|---|---|---|---|---|---|")
    for k in which:
        name, w, h, qp, keyint, frames, slices = CONFIGS[k]
        res = []
        for closed in ("0", "1"):   # the switch is read when the oracle opens: one process each
            env = dict(os.environ, RD_CHILD="1", H264O_CLOSED_LOOP_INTRA=closed)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), k], env=env, capture_output=True, text=True, check=True).stdout
            res.append([float(x) for x in out.split()])
        (b, bi, bp, py, pu, pv), (cb, cbi, cbp, cpy, _, _) = res
        print("| %s, %d frames pattern-B | %d | %.0f (%.0f / %.0f) | %.2f | %.2f / %.2f / %.2f | %.0f, %.2f | %+.1f %% bytes, %+.2f dB |" % (
            name, frames, qp, b, bi, bp, b * 8 * 30 / 1e6, py, pu, pv, cb, cpy, (b / cb - 1) * 100, py - cpy), flush=True)


if __name__ == "__main__":
    main()
