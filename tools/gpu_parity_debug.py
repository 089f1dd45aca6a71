#!/usr/bin/env python3
"""Stage-by-stage parity report of the HIP path against the CPU oracle (GPU box).

Development tool (not part of the product): for each case it encodes the same
synthetic frames with the oracle and with the HIP library and compares, per
frame, the padded source, the per-macroblock decisions, levels, coefficient
counts, the deblocked reconstruction and finally the Annex-B bytes. It keeps
going after a mismatch so one GPU run reports as much as possible.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol  # noqa: E402
import cedarx_h264_encoder_amd as ce  # noqa: E402

CASES = [
    # w, h, qp, keyint, frames, pattern, slices
    (64, 48, 30, 1, 2, "B", 1),
    (64, 48, 30, 25, 4, "B", 1),
    (64, 48, 24, 25, 3, "C", 1),
    (64, 48, 30, 25, 3, "A", 1),
    (70, 50, 24, 25, 3, "B", 1),
    (320, 240, 30, 25, 4, "B", 1),
    (320, 240, 24, 25, 3, "B", 4),
    (64, 48, 5, 3, 4, "C", 1),
    (64, 48, 47, 3, 4, "B", 1),
    (16, 16, 24, 2, 3, "C", 1),
    (854, 480, 30, 25, 3, "B", 1),
]
if len(sys.argv) > 1 and sys.argv[1] == "--quick":
    CASES = CASES[:4]


def first_diff(a, b):
    d = np.argwhere(a != b)
    return d[0].tolist() if len(d) else None


def cmp(name, a, b, log):
    if a.shape != b.shape:
        log("    %-12s SHAPE %s vs %s" % (name, a.shape, b.shape))
        return False
    n = int((a != b).sum())
    if n:
        i = first_diff(a, b)
        log("    %-12s MISMATCH %d/%d first@%s oracle=%s gpu=%s" % (
            name, n, a.size, i, a[tuple(i)], b[tuple(i)]))
        return False
    return True


def run_case(case, log):
    w, h, qp, keyint, frames, pat, slices = case
    log("CASE %s" % (case,))
    oc = ol.default_config(w, h, qp=qp, keyint=keyint, slice_count=slices)
    oe = ol.OracleEncoder(oc)
    ge = ce.CedarEncoder(width=w, height=h, qp=qp, keyframe_interval=keyint, slice_count=slices)
    dec = ol.OracleDecoder()
    ok_all = True
    for f in range(frames):
        y, uv = ol.gen_frame(pat, w, h, f)
        ref = oe.encode(y, uv)
        t0 = time.time()
        got = ge.encode_frame(y, uv)
        dt = (time.time() - t0) * 1e3
        same = ref == got
        log("  frame %d: oracle %d B, gpu %d B, %s (%.1f ms)" % (f, len(ref), len(got),
                                                               "BYTES OK" if same else "BYTES DIFFER", dt))
        om = oe.mb_array()
        gm = ge.debug_read("mbinfo")
        ok = True
        if f == 0:
            # padded source only needs one check
            pass
        for fld in ("type", "i16_mode", "chroma_mode", "cbp", "mvx", "mvy"):
            ok &= cmp("mb." + fld, om[fld], gm[fld], log)
        ok &= cmp("i4modes", om["i4_modes"], ge.debug_read("i4modes"), log)
        ok &= cmp("nnz", om["nnz"], ge.debug_read("nnz"), log)
        gc = ge.debug_read("coef")
        ok &= cmp("coef.luma", om["luma"].reshape(len(om), 256), gc[:, 0:256], log)
        ok &= cmp("coef.dc16", om["dc16"], gc[:, 256:272], log)
        ok &= cmp("coef.cdc", om["cdc"].reshape(len(om), 8), gc[:, 272:280], log)
        oac = om["cac"].reshape(len(om), 8, 16).copy()
        gac = gc[:, 280:408].reshape(len(om), 8, 16).copy()
        oac[:, :, 0] = 0
        gac[:, :, 0] = 0
        ok &= cmp("coef.cac", oac, gac, log)
        ry, ruv = oe.recon()
        ok &= cmp("recon_y", ry, ge.debug_read("recon_y"), log)
        ok &= cmp("recon_uv", ruv, ge.debug_read("recon_uv"), log)
        if not same:
            n = min(len(ref), len(got))
            d = next((i for i in range(n) if ref[i] != got[i]), n)
            log("    bytes first differ at %d: oracle %s gpu %s" % (d, ref[d:d + 8].hex(), got[d:d + 8].hex()))
        try:
            r = dec.decode(got)
            dy, duv = dec.picture()
            gy = ge.debug_read("recon_y")
            guv = ge.debug_read("recon_uv")
            if r != 1 or not (np.array_equal(dy, gy) and np.array_equal(duv, guv)):
                log("    decode(gpu stream) != gpu recon (r=%d, luma diff %d, chroma diff %d)" % (
                    r, int((dy != gy).sum()), int((duv != guv).sum())))
                ok = False
        except Exception as ex:  # noqa: BLE001
            log("    decoder rejected gpu stream: %s" % ex)
            ok = False
        ok_all &= ok and same
    ge.close()
    log("  => %s" % ("PASS" if ok_all else "FAIL"))
    return ok_all


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    fh = open(os.path.join(ROOT, "gpurun_out", "parity.log"), "w")

    def log(s):
        print(s, flush=True)
        fh.write(s + "\n")
        fh.flush()

    npass = 0
    for c in CASES:
        try:
            npass += bool(run_case(c, log))
        except Exception as ex:  # noqa: BLE001
            log("  EXCEPTION %r" % (ex,))
    log("SUMMARY %d/%d cases pass" % (npass, len(CASES)))
    return 0 if npass == len(CASES) else 1


if __name__ == "__main__":
    sys.exit(main())
